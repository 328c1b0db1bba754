// Whole-volume helpers either side of the patch path, gfx950: patch tiling + stitching of skull
// volumes (BASELINE config 4: "skull volumes tiled to 192^3 patches") and the Hausdorff-distance metric of the
// inference tail.  Index / integer work on NCDHW volumes: HBM-bound streaming; results are exact integers up to the
// final sqrt / division.
//
// Replaces: hausdorff (monai compute_hausdorff_distance on one_hot(argmax(pred)))  ctunet/utilities.py:62-70
//           (tiling has no counterpart in the reference: its datasets feed whole pre-resized volumes,
//            ctunet/pytorch/datasets.py:89-112,195-235; the tiles carry the same sample schema)
#include "common.h"

namespace {

constexpr int VB = 256;
constexpr int EDT_INF = 1 << 29;

// ------------------------------------------------------------------------------------------------ tiling
// out[p][c][z][y][x] = vol[c][z0+z][y0+y][x0+x], zero outside the volume (volumes smaller than a patch)
__global__ void extract_patches_kernel(const float* __restrict__ vol, const int32_t* __restrict__ coords, int P, int C, int D,
                                       int H, int W, int pd, int ph, int pw, float* __restrict__ out) {
    const int64_t pv = (int64_t)pd * ph * pw;
    const int64_t total = (int64_t)P * C * pv;
    for (int64_t idx = (int64_t)blockIdx.x * VB + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * VB) {
        const int64_t v = idx % pv;
        const int c = (int)((idx / pv) % C), p = (int)(idx / (pv * C));
        const int x = (int)(v % pw), y = (int)((v / pw) % ph), z = (int)(v / ((int64_t)pw * ph));
        const int gz = coords[p * 3] + z, gy = coords[p * 3 + 1] + y, gx = coords[p * 3 + 2] + x;
        float r = 0.f;
        if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) r = vol[(((int64_t)c * D + gz) * H + gy) * W + gx];
        out[idx] = r;
    }
}

// gather form (no atomics, fixed patch order => bitwise reproducible): out[c][v] = mean over the patches that cover v
__global__ void stitch_kernel(const float* __restrict__ patches, const int32_t* __restrict__ coords, int P, int C, int D, int H,
                              int W, int pd, int ph, int pw, float* __restrict__ out) {
    const int64_t V = (int64_t)D * H * W, pv = (int64_t)pd * ph * pw;
    const int64_t total = (int64_t)C * V;
    for (int64_t idx = (int64_t)blockIdx.x * VB + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * VB) {
        const int64_t v = idx % V;
        const int c = (int)(idx / V);
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((int64_t)W * H));
        float s = 0.f;
        int cnt = 0;
        for (int p = 0; p < P; ++p) {
            const int lz = z - coords[p * 3], ly = y - coords[p * 3 + 1], lx = x - coords[p * 3 + 2];
            if (lz >= 0 && lz < pd && ly >= 0 && ly < ph && lx >= 0 && lx < pw) {
                s += patches[((int64_t)p * C + c) * pv + ((int64_t)lz * ph + ly) * pw + lx];
                ++cnt;
            }
        }
        out[idx] = cnt ? s / (float)cnt : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ Hausdorff
// plane q = (n, class c >= 1, side): side 0 = hard segmentation argmax(pred) == c, side 1 = target[n][c] != 0
__device__ __forceinline__ bool seg_at(const float* pred, const float* target, int C, int64_t V, int n, int c, int side,
                                       int64_t v) {
    if (side) return target[((int64_t)n * C + c) * V + v] != 0.f;
    const float* p = pred + (int64_t)n * C * V;
    float best = p[v];
    int bi = 0;
    for (int k = 1; k < C; ++k) {
        const float xk = p[(int64_t)k * V + v];
        if (xk > best) { best = xk; bi = k; }
    }
    return bi == c;
}

// edge = mask & ~erode(mask) with the 6-neighbourhood and background outside the volume (scipy binary_erosion
// defaults, as monai get_mask_edges uses them); counts[q] += number of edge voxels
__global__ void edges_kernel(const float* __restrict__ pred, const float* __restrict__ target, int N, int C, int D, int H, int W,
                             uint8_t* __restrict__ edges, int* __restrict__ counts) {
    const int64_t V = (int64_t)D * H * W;
    const int q = blockIdx.y;                         // ((n * (C-1)) + (c-1)) * 2 + side
    const int side = q & 1, c = (q >> 1) % (C - 1) + 1, n = (q >> 1) / (C - 1);
    int local = 0;
    for (int64_t v = (int64_t)blockIdx.x * VB + threadIdx.x; v < V; v += (int64_t)gridDim.x * VB) {
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((int64_t)W * H));
        bool e = false;
        if (seg_at(pred, target, C, V, n, c, side, v)) {
            const bool inner = x > 0 && x < W - 1 && y > 0 && y < H - 1 && z > 0 && z < D - 1 &&
                               seg_at(pred, target, C, V, n, c, side, v - 1) && seg_at(pred, target, C, V, n, c, side, v + 1) &&
                               seg_at(pred, target, C, V, n, c, side, v - W) && seg_at(pred, target, C, V, n, c, side, v + W) &&
                               seg_at(pred, target, C, V, n, c, side, v - (int64_t)W * H) &&
                               seg_at(pred, target, C, V, n, c, side, v + (int64_t)W * H);
            e = !inner;
        }
        edges[(int64_t)q * V + v] = e ? 1 : 0;
        local += e;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counts[q], local);        // integer: order-independent
}

// exact squared Euclidean distance transform, separable min-plus passes (brute force along the line: the metric is not
// on the training path).  pass 0 along x from the edge bytes; passes 1, 2 along y, z from the previous pass.
__global__ void edt_pass_x_kernel(const uint8_t* __restrict__ edges, int64_t V, int W, int* __restrict__ g) {
    const int64_t v = (int64_t)blockIdx.x * VB + threadIdx.x;
    const int64_t base = (int64_t)blockIdx.y * V;
    if (v >= V) return;
    const int x = (int)(v % W);
    const uint8_t* row = edges + base + (v - x);
    int best = EDT_INF;
    for (int xx = 0; xx < W; ++xx)
        if (row[xx]) { const int dlt = x - xx; best = min(best, dlt * dlt); }
    g[base + v] = best;
}

// line along the axis with element stride `stride` and length L; `inner` = number of consecutive voxels that share a line
// index pattern (W for the y pass, W*H for the z pass)
__global__ void edt_pass_kernel(const int* __restrict__ src, int64_t V, int L, int64_t stride, int* __restrict__ dst) {
    const int64_t v = (int64_t)blockIdx.x * VB + threadIdx.x;
    const int64_t base = (int64_t)blockIdx.y * V;
    if (v >= V) return;
    const int pos = (int)((v / stride) % L);
    const int* line = src + base + (v - (int64_t)pos * stride);
    int best = EDT_INF;
    for (int k = 0; k < L; ++k) {
        const int dlt = pos - k;
        best = min(best, line[(int64_t)k * stride] + dlt * dlt);
    }
    dst[base + v] = best;
}

// maxd[q] = max over the edge voxels of plane q of the squared distance to the OTHER side's edges (plane q ^ 1)
__global__ void masked_max_kernel(const uint8_t* __restrict__ edges, const int* __restrict__ d2, int64_t V, int* __restrict__ maxd) {
    const int q = blockIdx.y;
    int local = -1;
    for (int64_t v = (int64_t)blockIdx.x * VB + threadIdx.x; v < V; v += (int64_t)gridDim.x * VB)
        if (edges[(int64_t)q * V + v]) local = max(local, d2[(int64_t)(q ^ 1) * V + v]);
    for (int o = 32; o > 0; o >>= 1) local = max(local, __shfl_xor(local, o));
    if ((threadIdx.x & 63) == 0 && local >= 0) atomicMax(&maxd[q], local);     // integer max: order-independent
}

__global__ void hausdorff_final_kernel(const int* __restrict__ counts, const int* __restrict__ maxd, int pairs, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs) return;
    const int a = 2 * i, b = 2 * i + 1;
    float r = __builtin_nanf("");                    // an empty surface on either side: monai yields nan / inf there
    if (counts[a] > 0 && counts[b] > 0) r = sqrtf((float)max(maxd[a], maxd[b]));
    out[i] = r;
}

}  // namespace

extern "C" int ctu_extract_patches(const float* vol, const int32_t* coords, int P, int C, int D, int H, int W, int pd, int ph,
                                   int pw, float* out, void* stream) {
    CTU_REQUIRE(vol && coords && out, "extract_patches: null pointer");
    CTU_REQUIRE(P > 0 && C > 0 && D > 0 && H > 0 && W > 0 && pd > 0 && ph > 0 && pw > 0, "extract_patches: bad shape");
    const int64_t total = (int64_t)P * C * pd * ph * pw;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(total, VB), 1 << 20);
    extract_patches_kernel<<<grid, VB, 0, (hipStream_t)stream>>>(vol, coords, P, C, D, H, W, pd, ph, pw, out);
    CTU_CHECK_LAUNCH("extract_patches");
    return CTU_OK;
}

extern "C" int ctu_stitch_patches(const float* patches, const int32_t* coords, int P, int C, int D, int H, int W, int pd, int ph,
                                  int pw, float* out, void* stream) {
    CTU_REQUIRE(patches && coords && out, "stitch_patches: null pointer");
    CTU_REQUIRE(P > 0 && C > 0 && D > 0 && H > 0 && W > 0 && pd > 0 && ph > 0 && pw > 0, "stitch_patches: bad shape");
    const int64_t total = (int64_t)C * D * H * W;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(total, VB), 1 << 20);
    stitch_kernel<<<grid, VB, 0, (hipStream_t)stream>>>(patches, coords, P, C, D, H, W, pd, ph, pw, out);
    CTU_CHECK_LAUNCH("stitch_patches");
    return CTU_OK;
}

// workspace: per plane (N * (C-1) * 2 of them) one edge byte map and two int32 distance maps, + 2 ints per plane
extern "C" size_t ctu_hausdorff_ws_bytes(int N, int C, int D, int H, int W) {
    if (N <= 0 || C < 2 || D <= 0 || H <= 0 || W <= 0) return 0;
    const size_t V = (size_t)D * H * W, Q = (size_t)N * (C - 1) * 2;
    return Q * V * 9 + 256 + Q * 2 * sizeof(int);
}

extern "C" int ctu_hausdorff(const float* pred, const float* target, int N, int C, int D, int H, int W, float* out, void* ws,
                             void* stream) {
    CTU_REQUIRE(pred && target && out && ws, "hausdorff: null pointer");
    CTU_REQUIRE(N > 0 && C >= 2 && C <= 8 && D > 0 && H > 0 && W > 0, "hausdorff: bad shape N=%d C=%d", N, C);
    CTU_REQUIRE(D <= 1024 && H <= 1024 && W <= 1024, "hausdorff: volume side above 1024");
    hipStream_t st = (hipStream_t)stream;
    const int64_t V = (int64_t)D * H * W;
    const int Q = N * (C - 1) * 2;
    CTU_REQUIRE(Q <= 65535, "hausdorff: too many (item, class) planes");
    int* d0 = (int*)ws;
    int* d1 = d0 + (size_t)Q * V;
    uint8_t* edges = (uint8_t*)(d1 + (size_t)Q * V);
    int* counts = (int*)(((uintptr_t)(edges + (size_t)Q * V) + 255) & ~(uintptr_t)255);
    int* maxd = counts + Q;
    if (hipMemsetAsync(counts, 0, sizeof(int) * 2 * Q, st) != hipSuccess) { ctu_set_error("hausdorff: memset failed"); return CTU_ELAUNCH; }
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(V, VB), 4096);
    edges_kernel<<<dim3(gx, Q), VB, 0, st>>>(pred, target, N, C, D, H, W, edges, counts);
    CTU_CHECK_LAUNCH("hausdorff edges");
    const dim3 full((unsigned)ceil_div64(V, VB), Q);
    edt_pass_x_kernel<<<full, VB, 0, st>>>(edges, V, W, d0);
    CTU_CHECK_LAUNCH("hausdorff edt x");
    edt_pass_kernel<<<full, VB, 0, st>>>(d0, V, H, (int64_t)W, d1);
    CTU_CHECK_LAUNCH("hausdorff edt y");
    edt_pass_kernel<<<full, VB, 0, st>>>(d1, V, D, (int64_t)W * H, d0);
    CTU_CHECK_LAUNCH("hausdorff edt z");
    masked_max_kernel<<<dim3(gx, Q), VB, 0, st>>>(edges, d0, V, maxd);
    CTU_CHECK_LAUNCH("hausdorff max");
    hausdorff_final_kernel<<<ceil_div(Q / 2, 64), 64, 0, st>>>(counts, maxd, Q / 2, out);
    CTU_CHECK_LAUNCH("hausdorff final");
    return CTU_OK;
}
