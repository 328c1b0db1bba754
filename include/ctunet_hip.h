/*
 * ctunet_hip.h -- C ABI of libctunet_hip.so: the MI355X (gfx950) kernels behind the
 * ctunet 3D U-Net hot path.
 *
 * The reference (vfmatzkin/ct-unet) has no FFI of its own: every device operation is a
 * stock torch.nn call inside ctunet/pytorch/models.py, utilities.py and ProblemHandler.py.
 * Each entry point below replaces one such call site (cited per function, paths relative
 * to the reference root).  The host side (ct-unet_amd/ctunet_amd) binds these with ctypes.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE
 *    pointers unless a parameter says "host".
 *  - Ownership: the library never allocates or frees tensor memory.  Outputs, saved
 *    statistics and workspaces are caller allocations.
 *  - Streams: kernels are enqueued on `stream` (a hipStream_t passed as void*) and the
 *    call returns without synchronising; calls are safe under stream capture.
 *  - Errors: return 0 on success, a negative CTU_E* code otherwise; ctu_last_error()
 *    returns a thread-local message for the last failing call on this thread.
 *  - Activations are fp32, channels-last-3d ("NDHWC") with a CHANNEL STRIDE `cs`
 *    (number of floats between consecutive voxels), so a tensor may be a channel slice
 *    of a wider buffer (the skip-concat buffers).  Channel counts `*_p` are padded to a
 *    multiple of 8; padded channels hold zeros.  Only ctu_ncdhw_to_ndhwc / ctu_head_*
 *    / ctu_loss_* touch the caller-visible NCDHW layout.
 *  - "Lazy BatchNorm": a conv writes its RAW output plus per-block sum / sum-of-squares
 *    partials; ctu_bn_finalize turns those into per-channel scale/shift; every consumer
 *    applies  a = relu(raw*scale + shift)  while loading (`in_scale`,`in_shift`,
 *    `in_relu`; NULL scale = identity).  Raw tensors are what backward re-reads.
 */
#ifndef CTUNET_HIP_H
#define CTUNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTU_OK 0
#define CTU_EINVAL (-1)   /* bad argument / unsupported shape */
#define CTU_ELAUNCH (-2)  /* HIP launch or runtime error       */

/* storage types of activation tensors: the fp32 entry points take float*; the ctu_lp_* (reduced precision) ones take a
 * dtype code and void* tensors of 16-bit elements (arithmetic stays fp32: MFMA accumulators, BatchNorm statistics,
 * weight gradients, master weights) */
#define CTU_F32 0
#define CTU_BF16 1
#define CTU_F16 2

/* In-launch BatchNorm finalize ("tail"), optional on every entry point that writes BatchNorm partial rows: with a
 * non-NULL tail the block of that launch that finishes last reduces the rows itself and writes what the separate
 * ctu_bn_finalize (forward rows) / ctu_bn_bwd_finalize (backward rows) launch would have written, with the same
 * arithmetic -- one launch less per BatchNorm layer and direction (34 per UNet() train step).  The fields are those
 * calls' arguments.  counter: one zero-initialised 32-bit word in device memory per layer and direction; the tail leaves
 * it zero, so a replayed graph needs no memset.  NULL tail = rows only (finalize with the separate call).
 * Reference: nn.BatchNorm3d in train mode and its autograd (ctunet/pytorch/models.py:27-32,39-44). */
typedef struct ctu_bn_tail {
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;          /* NULL: no running-statistics update */
    float* scale; float* shift; float* mean; float* invstd;
    long long* num_batches_tracked;                   /* NULL: none */
    unsigned int* counter;
    double count;                                     /* N*D*H*W of the normalised tensor */
    float momentum, eps;
    int C, n_updates;
} ctu_bn_tail;
typedef struct ctu_bn_bwd_tail {
    const float* gamma; const float* invstd;
    float* dgamma; float* dbeta; float* coef;         /* coef [5][cp] as ctu_bn_bwd_finalize */
    const float* mean; float* running_mean; float* running_var;   /* the replayed update; NULL running_mean: none */
    long long* num_batches_tracked;
    unsigned int* counter;
    double count;
    float momentum, eps;
    int C;
} ctu_bn_bwd_tail;

const char* ctu_last_error(void);
/* Library/ABI version (bumped on any signature change). */
int ctu_abi_version(void);
/* Name of the code object target this library was compiled for ("gfx950"). */
const char* ctu_arch(void);

/* ---------------------------------------------------------------- layout ---- */
/* NCDHW [N,C,D,H,W] -> NDHWC [N,D,H,W,cs], channels C..cp-1 zero-filled.
 * Host boundary of Model.forward_pass (ctunet/pytorch/Model.py:343-352). */
int ctu_ncdhw_to_ndhwc(const float* src, float* dst, int N, int C, int D, int H, int W,
                       int cp, int cs, void* stream);
/* NDHWC (channel stride cs) -> NCDHW, first C channels. */
int ctu_ndhwc_to_ncdhw(const float* src, float* dst, int N, int C, int D, int H, int W,
                       int cs, void* stream);

/* ------------------------------------------------------- conv3d (k=3 / k=5) ---- */
/* Packed-weight layouts.  0: [8-ch chunk][tap][16-wide out tile][kq][n][j].  1 ("pair", only k = 3 with
 * nout_p = 8): the MFMA N axis carries (w-shift, out channel) and M carries voxel pairs, 36 taps with
 * zero-filled entries -- 1.5x fewer MFMAs than padding 8 channels to 16.  ctu_conv3d_layout returns the
 * layout the forward kernel is fastest with for a volume of width W; pack and forward must agree.
 * Geometry helpers: packed-weight size (floats) and number of spatial blocks (= rows of the
 * stats-partials buffer) of a conv call. */
int ctu_conv3d_layout(int k, int nout_p, int W);
/* Name of the device kernel a ctu_conv3d_fwd / ctu_conv3d_wgrad call with this geometry launches (as it
 * appears in a rocprofv3 kernel trace; thread-local string) -- used to match timings with profiles. */
const char* ctu_conv3d_fwd_kernel_name(int N, int D, int H, int W, int k, int nout_p, int layout);
const char* ctu_conv3d_wgrad_kernel_name(int W, int k, int cin_p, int cout_p);
size_t ctu_conv3d_packed_floats(int k, int rin_p, int nout_p, int layout);
int ctu_conv3d_num_blocks(int N, int D, int H, int W, int k, int nout_p, int layout);

/* Re-layout a torch Conv3d weight [Co,Ci,k,k,k] for the implicit-GEMM kernels.
 *  mode 0 (forward): reduction side = padded input channels, output side = co.
 *  mode 1 (data gradient): reduction side = co, output side = padded input channels, taps flipped.
 * Channel maps (int32, device, NULL = identity) come in two directions throughout this header:
 *   imap[logical channel]   -> position in the padded channels-last buffer
 *   cinv[padded position]   -> logical channel, -1 for a padding slot
 * (the concat buffers hold [C real | pad | C real | pad]).  rin_p / nout_p: padded channel
 * counts of the reduction and output sides of THIS packing (multiples of 8).  Every element of
 * wp is written (padding slots get 0), no prior memset needed.
 * nn.Conv3d weights: ctunet/pytorch/models.py:26,29,38,41,71,76,403,407,430,434,482-488. */
int ctu_pack_conv3d_weight(const float* w, float* wp, int Co, int Ci, int k,
                           const int32_t* cinv, int rin_p, int nout_p, int mode, int layout,
                           void* stream);

/* Every weight tensor of a network in ONE launch (the job table travels in the kernel arguments).
 * jobs: HOST array; kind 0 = Conv3d (ctu_pack_conv3d_weight semantics: w [Co,Ci,k,k,k], k, layout),
 * kind 1 = ConvTranspose3d (ctu_pack_convt_weight semantics: w [Ci,Co,2,2,2]; k/layout ignored). */
typedef struct {
    const float* w;
    float* wp;
    const int32_t* cinv;
    int kind, Co, Ci, k, rin_p, nout_p, mode, layout;
} ctu_pack_job;
int ctu_pack_batch(const ctu_pack_job* jobs, int n, void* stream);

/* Implicit-GEMM 3D convolution on MFMA (v_mfma_f32_16x16x4_f32), stride 1, zero padding
 * (k-1)/2, NDHWC.  out[v, o] = bias[o] + sum_{tap,r} A(in[v+tap, r]) * wp[tap, r, o] with
 * A() the lazy-BN input transform.  Used for nn.Conv3d forward (models.py call sites
 * above) and, with a mode-1 packing, for its data gradient.
 *  in/in_cs/rin_p ...... input tensor, channel stride, channels to contract (mult. of 8)
 *  in_scale/in_shift ... per-channel transform of `in` (NULL = identity); in_relu 0/1
 *  bias/nbias .......... [nbias] (logical, unpadded) or NULL
 *  out/out_cs/nout_p ... output tensor (channel slice base), stride, channels written
 *  stats ............... NULL or [ctu_conv3d_num_blocks()][2][nout_p]: per-block sum and
 *                        sum of squares of the written output (train-mode BatchNorm3d,
 *                        models.py:27,31,39,43,74,79,405,409,432,436,485,490) */
int ctu_conv3d_fwd(const float* in, int in_cs, int rin_p,
                   const float* in_scale, const float* in_shift, int in_relu,
                   const float* wp, const float* bias, int nbias,
                   float* out, int out_cs, int nout_p, float* stats,
                   int N, int D, int H, int W, int k, int layout, const ctu_bn_tail* tail, void* stream);

/* Weight gradient of nn.Conv3d: dW[co,ci,tap] = sum_v A(in[v+tap, pos(ci)]) * gout[v, co],
 * pos = inverse of cinv.
 * ws: workspace of ctu_conv3d_wgrad_ws_floats() floats (per-wave partial slabs, reduced
 * deterministically by a second kernel; no float atomics).  dw: torch layout [Co,Ci,k,k,k].
 * dbias: NULL or [Co] = sum_v gout[v,co]. */
size_t ctu_conv3d_wgrad_ws_floats(int N, int D, int H, int W, int k, int cin_p, int cout_p);
int ctu_conv3d_wgrad(const float* in, int in_cs, int cin_p,
                     const float* in_scale, const float* in_shift, int in_relu,
                     const float* gout, int g_cs, int cout_p,
                     float* dw, float* dbias, int Co, int Ci, const int32_t* cinv,
                     float* ws, int N, int D, int H, int W, int k, void* stream);
/* The same with the BatchNorm3d + ReLU backward of the layer (models.py:27-32, torch autograd) folded in: ga is the
 * gradient w.r.t. the ACTIVATED output, y the layer's raw output (same geometry and channel stride g_cs as ga), bn_scale /
 * bn_shift its ctu_bn_finalize vectors, coef the [5][cout_p] rows of ctu_bn_bwd_finalize.  The kernel forms the raw-output
 * gradient gy while staging ga, uses it for dW and writes it to gy_out (geometry of ga, must not alias it) for the
 * data-gradient call that follows -- the ctu_bn_relu_bwd_apply pass over the layer disappears.  Only for the geometries
 * ctu_conv3d_wgrad_bn_supported accepts (k = 3, volume a multiple of the 4 x 4 x 8|16 box, full channel tiles). */
int ctu_conv3d_wgrad_bn_supported(int N, int D, int H, int W, int k, int cin_p, int cout_p);
int ctu_conv3d_wgrad_bn(const float* in, int in_cs, int cin_p,
                        const float* in_scale, const float* in_shift, int in_relu,
                        const float* ga, int g_cs, int cout_p, const float* y,
                        const float* bn_scale, const float* bn_shift, const float* coef, float* gy_out,
                        float* dw, int Co, int Ci, const int32_t* cinv,
                        float* ws, int N, int D, int H, int W, int k, void* stream);

/* First encoder convolution: C_in = 1 or 2, k = 3, at most 8 output channels (nn.Conv3d(input_channels, i_size, 3),
 * models.py:26 with the channel plan of :175,272-296).  K = 27*C_in is too short for the implicit-GEMM tile and the
 * layer is HBM-bound, so it gets direct kernels that read / write the caller's NCDHW planes in place (no padded
 * channels-last copy of the input):
 *   fwd ....... x [N,cin,D,H,W] (NCDHW) * w [Co,cin,3,3,3] (torch layout, unpacked) -> out channels-last (8 padded
 *               channels, stride out_cs) + one BN partial row [2][8] per block (ctu_conv3d_first_num_blocks rows)
 *   bwd_data .. g channels-last (8 padded channels) -> dx [N,cin,D,H,W] (NCDHW)
 *   wgrad ..... dw [Co,cin,3,3,3]; ws: ctu_conv3d_first_wgrad_ws_floats() floats
 * ctu_conv3d_first_supported tells whether a layer qualifies (k = 3, cin <= 2, nout_p = 8, W >= 16). */
int ctu_conv3d_first_supported(int k, int cin, int nout_p, int W);
int ctu_conv3d_first_num_blocks(int N, int D, int H, int W);
int ctu_conv3d_first_fwd(const float* x, int cin, const float* w, const float* bias, int nbias,
                         float* out, int out_cs, int Co, float* stats,
                         int N, int D, int H, int W, const ctu_bn_tail* tail, void* stream);
int ctu_conv3d_first_bwd_data(const float* g, int g_cs, const float* w, int cin, int Co, float* dx,
                              int N, int D, int H, int W, void* stream);
size_t ctu_conv3d_first_wgrad_ws_floats(int N, int D, int H, int W, int cin);
int ctu_conv3d_first_wgrad(const float* x, int cin, const float* g, int g_cs, float* dw, int Co,
                           float* ws, int N, int D, int H, int W, void* stream);
/* ctu_conv3d_first_wgrad with the layer's BatchNorm + ReLU backward folded in (see ctu_conv3d_wgrad_bn): ga = gradient
 * w.r.t. the ACTIVATED output, y = the raw output, coef = ctu_bn_bwd_finalize's [5][8] rows; the raw-output gradient is
 * written to gy_out (geometry of ga) for ctu_conv3d_first_bwd_data. */
int ctu_conv3d_first_wgrad_bn(const float* x, int cin, const float* ga, int g_cs, const float* y, const float* bn_scale,
                              const float* bn_shift, const float* coef, float* gy_out, float* dw, int Co, float* ws,
                              int N, int D, int H, int W, void* stream);

/* ------------------------------------------------------------ BatchNorm3d ---- */
/* Train mode: reduce the per-block partials written by ctu_conv3d_fwd into batch
 * statistics and the lazy transform.  count = N*D*H*W.  C logical channels, cp padded.
 *  scale = gamma*invstd, shift = beta - mean*scale (0 for padded channels)
 *  running_mean/var updated `n_updates` times with momentum (unbiased var), as
 *  nn.BatchNorm3d does; n_updates = 2 reproduces the double update that
 *  torch.utils.checkpoint causes (models.py:232-255).
 * mean_out/invstd_out [cp] are saved for backward. */
int ctu_bn_finalize(const float* stats, int nblocks, int C, int cp, double count,
                    const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    int n_updates, float* scale, float* shift, float* mean_out,
                    float* invstd_out, long long* num_batches_tracked, void* stream);
/* (num_batches_tracked: the BatchNorm's int64 counter on the device, += n_updates by the same launch; NULL = none.
 *  ctu_bn_bwd_finalize takes it too and adds 1 when it replays the running-stat update.) */
/* Eval mode: scale/shift from the running statistics. */
int ctu_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, int cp,
                       float* scale, float* shift, void* stream);

/* Backward of  a = relu(bn(y))  for one conv output.
 * Pass 1: partials[nb][2][cp] of  sum(gz)  and  sum(gz * yhat),  gz = ga * (a > 0).
 * Pass 2 (after ctu_bn_bwd_finalize): gy = gamma*invstd*(gz - dbeta/n - yhat*dgamma/n),
 * written IN PLACE over ga.  y/ga are channels-last with strides.  */
int ctu_bn_bwd_num_blocks(int64_t nvox);
int ctu_bn_relu_bwd_reduce(const float* y, int y_cs, const float* ga, int g_cs, int cp,
                           const float* scale, const float* shift, const float* mean,
                           const float* invstd, int64_t nvox, float* partials, const ctu_bn_bwd_tail* tail, void* stream);
/* coef [5][cp]: rows k0 = gamma*invstd, k1 = dbeta/n, k2 = dgamma/n (what ctu_bn_relu_bwd_apply reads) and the same
 * backward as one affine map of the raw output, A = -k0*k2*invstd, B = -k0*(k1 - k2*mean*invstd):
 *     gy = (y*scale + shift > 0 ? k0*ga : 0) + A*y + B
 * (what the weight-gradient kernels of ctu_conv3d_wgrad_bn / ctu_upconv_fused_wgrad_bn apply while staging ga).
 * running_mean/running_var non-NULL: additionally replay the running-statistics update once, with the batch
 * statistics saved by ctu_bn_finalize (mean, invstd) -- the second update torch.utils.checkpoint's recompute
 * performs in backward when use_checkpoint=True (models.py:232-255). */
int ctu_bn_bwd_finalize(const float* partials, int nb, int C, int cp, double count,
                        const float* gamma, const float* invstd,
                        float* dgamma, float* dbeta, float* coef, const float* mean,
                        float* running_mean, float* running_var, float momentum, float eps,
                        long long* num_batches_tracked, void* stream);
int ctu_bn_relu_bwd_apply(const float* y, int y_cs, float* ga, int g_cs, int cp,
                          const float* scale, const float* shift, const float* mean,
                          const float* invstd, const float* coef, int64_t nvox, void* stream);

/* -------------------------------------------------------------- MaxPool3d ---- */
/* nn.MaxPool3d(2, stride 2) (models.py:190-191,233,469-470) of a = A(in); writes the
 * pooled ACTIVATED values (no transform needed downstream). */
int ctu_maxpool2_fwd(const float* in, int in_cs, int cp, const float* in_scale,
                     const float* in_shift, int in_relu, float* out, int out_cs,
                     int N, int D, int H, int W, void* stream);
/* Backward: routes gout (pooled grid) to the first max of each 2x2x2 window of A(in)
 * (recomputed, no stored indices) and ADDS it into gin (full grid, stride gin_cs). */
int ctu_maxpool2_bwd(const float* in, int in_cs, int cp, const float* in_scale,
                     const float* in_shift, int in_relu, const float* gout, int gout_cs,
                     float* gin, int gin_cs, int accumulate, int N, int D, int H, int W,
                     void* stream);
/* The same, when in is the raw output of a conv whose train-mode BatchNorm + ReLU (in_scale/in_shift, mean/invstd) the
 * pooled tensor is, and gin is complete after this call (the reference's encoder: pool(block(x)), models.py:233-246,
 * with the skip's share already in gin): also emits that BatchNorm's backward reduction -- one row {sum gz[c], sum
 * gz[c]*xhat[c]} per block, ctu_maxpool2_bwd_bn_num_blocks rows of 2*cp floats, consumed by ctu_bn_bwd_finalize in place
 * of ctu_bn_relu_bwd_reduce's rows (saves one pass over y and the gradient).  ctu_maxpool2_bwd_bn_num_blocks returns 0
 * for channel counts the fused form does not take (cp/4 must divide 256): use ctu_maxpool2_bwd + the separate reduce. */
int ctu_maxpool2_bwd_bn_num_blocks(int N, int D, int H, int W, int cp);
int ctu_maxpool2_bwd_bn(const float* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                        const float* mean, const float* invstd, const float* gout, int gout_cs, float* gin,
                        int gin_cs, int accumulate, int N, int D, int H, int W, float* partials, const ctu_bn_bwd_tail* tail, void* stream);

/* -------------------------------------------------- ConvTranspose3d k2 s2 ---- */
/* nn.ConvTranspose3d(C, C, 2, 2) with bias (models.py:37,427-429):
 * out[2v+tap, o] = b[o] + sum_r A(in[v, r]) * w[r, o, tap];  w torch layout [Ci,Co,2,2,2].
 * Packing: mode 0 forward (reduction = padded input channels, output = co),
 *          mode 1 data gradient (reduction = co, output = padded input channels); cinv as above. */
size_t ctu_convt_packed_floats(int rin_p, int nout_p);
int ctu_pack_convt_weight(const float* w, float* wp, int Ci, int Co, const int32_t* cinv,
                          int rin_p, int nout_p, int mode, void* stream);
int ctu_convt2_fwd(const float* in, int in_cs, int rin_p, const float* in_scale,
                   const float* in_shift, int in_relu, const float* wp, const float* bias,
                   int nbias, float* out, int out_cs, int nout_p, int N, int D, int H, int W,
                   void* stream);      /* D,H,W: INPUT grid; output grid is 2D,2H,2W */
/* gin[v, r] = sum_{tap,o} gout[2v+tap, o] * w[r, o, tap]  (wp from mode 1). */
int ctu_convt2_bwd_data(const float* gout, int g_cs, int rout_p, const float* wp,
                        float* gin, int gin_cs, int nin_p, int N, int D, int H, int W,
                        void* stream);
/* dw[ci,co,tap] = sum_v A(in[v, imap[ci]]) * gout[2v+tap, co];  dbias[co] = sum gout. */
size_t ctu_convt2_wgrad_ws_floats(int N, int D, int H, int W, int cin_p, int cout_p);
int ctu_convt2_wgrad(const float* in, int in_cs, int cin_p, const float* in_scale,
                     const float* in_shift, int in_relu, const float* gout, int g_cs,
                     int cout_p, float* dw, float* dbias, int Ci, int Co,
                     const int32_t* imap, float* ws, int N, int D, int H, int W, void* stream);

/* ------------------------------------------------------------------- head ---- */
/* last_conv 1x1x1 + bias, optional softmax(dim=1), optional sigmoid, optional SP
 * re-encoding, written NCDHW (models.py:223-224,255-259,317-330,364-365,507,538).
 *  w [Co,Ci] torch layout, imap as above, act: bit0 softmax, bit1 sigmoid.  Co <= 4.
 *  head_mode 0: out0 = y [N,Co,...].
 *  head_mode 1 (UNetSP/UNetDO): out0 = [y0, y1+y2], out1 = [1-y1, y1]  (Co must be 3).
 *  head_mode 2 (UNetSPSmall): mode 1 followed by softmax of each pair. */
int ctu_head_fwd(const float* in, int in_cs, int cin_p, const float* in_scale,
                 const float* in_shift, int in_relu, const float* w, const float* bias,
                 const int32_t* imap, int Ci, int Co, int act, int head_mode,
                 float* out0, float* out1, int N, int64_t nvox_per_item, void* stream);
/* Backward of the head; the forward values are recomputed from `in` (nothing saved).
 * g0/g1: gradients of out0/out1 (NCDHW, g1 NULL for mode 0).  Produces gin (channels-last,
 * stride gin_cs, cin_p channels = gradient w.r.t. the ACTIVATED input), dw [Co,Ci], db [Co].
 * ws: workspace of ctu_head_bwd_ws_floats() floats. */
size_t ctu_head_bwd_ws_floats(int N, int64_t nvox_per_item, int cin_p, int Co);
int ctu_head_bwd(const float* in, int in_cs, int cin_p, const float* in_scale,
                 const float* in_shift, int in_relu, const float* w, const float* bias,
                 const int32_t* imap, int Ci, int Co, int act, int head_mode,
                 const float* g0, const float* g1, float* gin, int gin_cs,
                 float* dw, float* db, float* ws, int N, int64_t nvox_per_item, void* stream);
/* The same, when the first bn_cp input channels are the train-mode BatchNorm + ReLU output of one conv (in_scale/in_shift,
 * bn_mean/bn_invstd) and nothing else consumes that tensor (the decoder's last block feeding the head, models.py:252-255):
 * also emits that BatchNorm's backward reduction, ctu_head_bwd_num_blocks rows of 2*bn_cp floats {sum gz, sum gz*xhat},
 * consumed by ctu_bn_bwd_finalize in place of ctu_bn_relu_bwd_reduce's rows. */
int ctu_head_bwd_num_blocks(int N, int64_t nvox_per_item);
int ctu_head_bwd_bn(const float* in, int in_cs, int cin_p, const float* in_scale,
                    const float* in_shift, int in_relu, const float* w, const float* bias,
                    const int32_t* imap, int Ci, int Co, int act, int head_mode,
                    const float* g0, const float* g1, float* gin, int gin_cs,
                    float* dw, float* db, float* ws, int N, int64_t nvox_per_item,
                    const float* bn_mean, const float* bn_invstd, int bn_cp, float* bn_partials, const ctu_bn_bwd_tail* tail, void* stream);

/* ------------------------------------------------------------------- loss ---- */
/* Fused Dice + cross-entropy on one 2-channel NCDHW map (utilities.py:35-50,
 * ProblemHandler.py:59-88,228-298).  pred/target [N,2,V].
 *  ce term   = ce_lambda   * CrossEntropy(pred as logits, argmax(target,1)), mean over N*V
 *  dice term = dice_lambda * dice_loss(P, target), P = softmax(pred,1) if dice_softmax else pred
 * terms (device, float[2]) = {ce term, dice term}.  ws: ctu_loss_ws_floats(N, V) floats,
 * kept until ctu_loss_bwd, which writes
 *   gpred = gscale_ce[0] * d(ce term)/dpred + gscale_dice[0] * d(dice term)/dpred
 * (gscale_*: one device float each -- the two upstream gradients autograd hands over, read in place; NULL = 1;
 *  accumulate != 0: gpred += ). */
size_t ctu_loss_ws_floats(int N, int64_t V);
int ctu_loss_fwd(const float* pred, const float* target, int N, int64_t V, float ce_lambda,
                 float dice_lambda, int dice_softmax, float* terms, float* ws, void* stream);
int ctu_loss_bwd(const float* pred, const float* target, int N, int64_t V, float ce_lambda,
                 float dice_lambda, int dice_softmax, const float* ws, const float* gscale_ce,
                 const float* gscale_dice, float* gpred, int accumulate, void* stream);

/* -------------------------------------------------------------- utilities ---- */
/* Additive skip connection, UNet(cat=False): out = act_a(a) + act_b(b) on channels-last tensors, where act_x is the
 * lazy BatchNorm(+ReLU) transform of that operand (scale/shift NULL: identity).  b == NULL: out = act_a(a), which is
 * also the strided channel-slice copy the backward of the add uses (/root/reference/ctunet/pytorch/models.py:250-251). */
int ctu_skip_add(const float* a, int a_cs, const float* a_scale, const float* a_shift, int a_relu,
                 const float* b, int b_cs, const float* b_scale, const float* b_shift, int b_relu,
                 float* out, int out_cs, int cp, int64_t nvox, void* stream);
/* ------------------------------------------- fused up-convolution (decoder) ---- */
/* ConvTranspose3d(C, C, 2, 2, bias) followed by Conv3d(C, Co, 3, padding 1, no bias) -- the first two layers of
 * every decoder block (/root/reference/ctunet/pytorch/models.py:37-38) -- as ONE convolution on the coarse grid:
 * per output parity a 2x2x2 convolution with composite weights W_eff = WT o W3 and a bias that only differs on the
 * volume faces (27 border classes).  Same result as the two layers up to fp32 summation order; 8 taps instead of
 * 27 + the transposed conv, and the fine-grid intermediate is never materialised.
 *   ctu_upconv_fused_pack: wt [C][C][2][2][2], bt [C], w3 [Co][C][3][3][3] (torch layouts) -> wp (packed_floats)
 *     and beff [27][nout_p]; cinv maps padded input positions to logical channels (concat layout), NULL = identity;
 *     ws = pack_ws_floats scratch (transposed copies of the two weight tensors).
 *   ctu_upconv_fused_fwd: in = COARSE activations [N,D,H,W,in_cs] with the lazy BatchNorm transform, out = RAW
 *     fine-grid output [N,2D,2H,2W,out_cs]; stats = num_blocks rows of [2][nout_p] partial sums for ctu_bn_finalize. */
int ctu_upconv_fused_supported(int k, int D, int H, int W, int cin_p, int nout_p);
size_t ctu_upconv_fused_packed_floats(int cin_p, int nout_p);
int ctu_upconv_fused_num_blocks(int N, int D, int H, int W, int nout_p);
size_t ctu_upconv_fused_pack_ws_floats(int C, int nout_p);
int ctu_upconv_fused_pack(const float* wt, const float* bt, const float* w3, int C, int Co, const int32_t* cinv,
                          int cin_p, int nout_p, float* wp, float* beff, float* ws, void* stream);
int ctu_upconv_fused_fwd(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                         int in_relu, const float* wp, const float* beff, float* out, int out_cs, int nout_p,
                         float* stats, int N, int D, int H, int W, const ctu_bn_tail* tail, void* stream);

/* Backward of the fused up-convolution w.r.t. the three parameter tensors (its data gradient: ctu_upconv_fused_bwd_data
 * below).  ctu_upconv_fused_wgrad: dweff [8 parities][8 taps][cin_p][nout_p] = sum_i x[i+d]^T dy[2i+p]
 * (in = COARSE activations with the lazy transform, gout = fine-grid gradient of the fused op's raw output).
 * ctu_upconv_fused_project: dWT, dbT, dW3 (torch layouts) from dweff and the border-aware sums of gout; pack_ws is
 * the scratch ctu_upconv_fused_pack filled in this step's forward, imap maps logical input channels to padded positions. */
size_t ctu_upconv_fused_wgrad_ws_floats(int N, int D, int H, int W, int cin_p, int nout_p);
int ctu_upconv_fused_wgrad(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                           int in_relu, const float* gout, int g_cs, int nout_p, float* dweff, float* ws,
                           int N, int D, int H, int W, void* stream);
/* ctu_upconv_fused_wgrad with the BatchNorm + ReLU backward of the fused op's output folded in (as ctu_conv3d_wgrad_bn):
 * ga = fine-grid gradient w.r.t. the ACTIVATED output, y = the fused op's raw output, gy_out = the raw-output gradient the
 * ctu_upconv_fused_project and ctu_upconv_fused_bwd_data calls that follow read.  N, D, H, W: COARSE dims. */
int ctu_upconv_fused_wgrad_bn_supported(int N, int D, int H, int W, int cin_p, int nout_p);
int ctu_upconv_fused_wgrad_bn(const float* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                              int in_relu, const float* ga, int g_cs, int nout_p, const float* y,
                              const float* bn_scale, const float* bn_shift, const float* coef, float* gy_out,
                              float* dweff, float* ws, int N, int D, int H, int W, void* stream);
size_t ctu_upconv_fused_project_ws_floats(int nout_p, int64_t fine_nvox);
int ctu_upconv_fused_project(const float* dweff, const float* gout, int g_cs, int nout_p, int N, int D, int H, int W,
                             const float* bt, const float* pack_ws, const int32_t* imap, int C, int Co, int cin_p,
                             float* dwt, float* dbt, float* dw3, float* ws, void* stream);

/* Data gradient of the fused up-convolution: gin (COARSE, [N,D,H,W,gin_cs], cin_p padded channels in the input's
 * layout) from gout (fine grid, gradient of the raw output) -- the adjoint parity convolutions, no fine-grid
 * intermediate.  wpd = ctu_upconv_fused_pack_bwd(wp): the forward packing re-gathered for the transposed product. */
size_t ctu_upconv_fused_bwd_packed_floats(int cin_p, int nout_p);
int ctu_upconv_fused_pack_bwd(const float* wp, int cin_p, int nout_p, float* wpd, void* stream);
int ctu_upconv_fused_bwd_data(const float* gout, int g_cs, int nout_p, const float* wpd, float* gin, int gin_cs,
                              int cin_p, int N, int D, int H, int W, void* stream);

/* -------------------------------------------------- inference tail / sample schema ---- */
/* hard_segm_from_tensor (/root/reference/ctunet/utilities.py:103-124): seg[n,v] = (float)argmax_c prob[n,c,v] over an
 * NCDHW map; the first maximum wins (torch.argmax). */
int ctu_hard_segm(const float* prob, int N, int C, int64_t nvox_per_item, float* seg, void* stream);
/* one_hot(label.long(), C).movedim(-1, 1).float() of the reference datasets (ctunet/pytorch/datasets.py:107-110,
 * 212-217): label [N,V] float -> out [N,C,V].  Labels outside [0,C) give an all-zero voxel (torch raises). */
int ctu_one_hot(const float* label, int N, int C, int64_t nvox_per_item, float* out, void* stream);
/* dice_coeff (ctunet/utilities.py:53-59 = monai compute_meandice on one_hot(argmax(pred,1))): counts[n][c] =
 * { sum hard_c*target_c, sum hard_c, sum target_c } in double (exact integers for one-hot targets), two-stage
 * fixed-order reduction.  C <= 8.  The Dice ratio itself (and the empty-vs-empty convention) is the caller's. */
size_t ctu_hard_dice_ws_doubles(int N);
int ctu_hard_dice_counts(const float* pred, const float* target, int N, int C, int64_t nvox_per_item,
                         double* counts, double* ws, void* stream);

/* hausdorff (ctunet/utilities.py:62-70 = monai compute_hausdorff_distance on one_hot(argmax(pred,1)), background
 * excluded, Euclidean, symmetric, no percentile): out[n][c-1], c = 1..C-1 = max over the surface voxels of either set of
 * the distance to the other set's surface; surface = mask & ~erode(mask) (6-neighbourhood, background outside the
 * volume); the distances come from an exact integer squared-distance transform.  NaN where either surface is empty (the
 * caller maps it, utilities.py:69).  pred / target [N,C,D,H,W]; ws: ctu_hausdorff_ws_bytes() bytes.  PARITY UNPINNED
 * (monai is not in the reference tree); tested against the same definition on scipy.ndimage. */
size_t ctu_hausdorff_ws_bytes(int N, int C, int D, int H, int W);
int ctu_hausdorff(const float* pred, const float* target, int N, int C, int D, int H, int W, float* out, void* ws,
                  void* stream);

/* Patch tiling of whole volumes (BASELINE config 4: skull volumes tiled to 192^3 patches; the tiles carry the sample
 * schema of ctunet/pytorch/datasets.py:89-112,195-235).  coords: DEVICE int32 [P][3] = (z0, y0, x0) of each patch.
 *   extract: out [P,C,pd,ph,pw] = vol [C,D,H,W] windows, zero-filled outside the volume
 *   stitch:  out [C,D,H,W] = mean over the patches covering each voxel (fixed patch order, no atomics), 0 if none */
int ctu_extract_patches(const float* vol, const int32_t* coords, int P, int C, int D, int H, int W, int pd, int ph,
                        int pw, float* out, void* stream);
int ctu_stitch_patches(const float* patches, const int32_t* coords, int P, int C, int D, int H, int W, int pd, int ph,
                       int pw, float* out, void* stream);

/* ------------------------------------------------ reduced precision (bf16 / fp16 activations) ---- */
/* BASELINE configs 4 ("bf16, 192^3 patches") and 5 ("fp16 MFMA conv path, 256^3 patches").  Same operations, call sites
 * and argument meaning as the fp32 entry points above, with
 *   dtype ............ CTU_BF16 | CTU_F16: element type of every ACTIVATION / activation-gradient tensor (void*), still
 *                      channels-last with channel counts padded to 8 and channel strides in ELEMENTS (multiples of 8)
 *   arithmetic ....... fp32: MFMA accumulators (v_mfma_f32_16x16x32_{bf16,f16}), lazy BatchNorm transform, statistics
 *                      (taken from the ROUNDED outputs), weight / bias / BatchNorm gradients, master weights
 *   packed weights ... 16-bit copies in MFMA fragment order, re-packed from the fp32 masters (ctu_lp_pack_*)
 * fp16 gradients need the caller's loss scaling (the per-voxel loss gradient of a 256^3 patch is 6e-8); bf16 does not. */
/* packed-weight layouts: 0 = [K-step][16-wide out tile][lane][8]; 1 ("pair": k = 3, 8 padded channels on both sides, W >= 32) =
 * (w-shift, channel) rows, a K-step = the 4 w offsets of one (kd, kh) row.  ctu_lp_conv3d_layout returns the layout the forward /
 * data-gradient kernel wants; pack, num_blocks and forward must be given the same one. */
int ctu_lp_conv3d_layout(int k, int rin_p, int nout_p, int W);
size_t ctu_lp_conv3d_packed_elems(int k, int rin_p, int nout_p);
int ctu_lp_conv3d_num_blocks(int N, int D, int H, int W, int k, int rin_p, int nout_p, int layout);
int ctu_lp_pack_conv3d_weight(int dtype, const float* w, void* wp, int Co, int Ci, int k, const int32_t* cinv,
                              int rin_p, int nout_p, int mode, int layout, void* stream);
/* every 16-bit weight copy of a network in ONE launch (ctu_pack_job as above; `layout` unused, kind 1 = ConvTranspose3d) */
int ctu_lp_pack_batch(int dtype, const ctu_pack_job* jobs, int n, void* stream);
/* nn.Conv3d forward (mode-0 packing) / data gradient (mode-1 packing); stats: [ctu_lp_conv3d_num_blocks()][2][nout_p]
 * (the voxel box of a launch grows when rin_p is small, so the row count depends on it) */
int ctu_lp_conv3d_fwd(int dtype, const void* in, int in_cs, int rin_p, const float* in_scale, const float* in_shift,
                      int in_relu, const void* wp, const float* bias, int nbias, void* out, int out_cs, int nout_p,
                      float* stats, int N, int D, int H, int W, int k, int layout, const ctu_bn_tail* tail, void* stream);
/* weight gradient -> dw fp32 [Co,Ci,k,k,k] (torch layout); ws: ctu_lp_conv3d_wgrad_ws_floats() floats */
size_t ctu_lp_conv3d_wgrad_ws_floats(int N, int D, int H, int W, int k, int cin_p, int cout_p);
int ctu_lp_conv3d_wgrad(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                        int in_relu, const void* gout, int g_cs, int cout_p, float* dw, int Co, int Ci,
                        const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, void* stream);
/* ctu_conv3d_wgrad_bn for 16-bit tensors (BatchNorm + ReLU backward folded into the weight-gradient kernel's staging; the
 * raw-output gradient is rounded once to the storage type and written to gy_out): k = 3, volumes at least 16 wide that are
 * multiples of the 4 x 4*(32/bw) x bw box (ctu_lp_conv3d_wgrad_bn_supported). */
int ctu_lp_conv3d_wgrad_bn_supported(int N, int D, int H, int W, int k, int cin_p, int cout_p);
int ctu_lp_conv3d_wgrad_bn(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                           int in_relu, const void* ga, int g_cs, int cout_p, const void* y, const float* bn_scale,
                           const float* bn_shift, const float* coef, void* gy_out, float* dw, int Co, int Ci,
                           const int32_t* cinv, float* ws, int N, int D, int H, int W, int k, void* stream);

/* first encoder convolution (C_in <= 2): the input x and dx stay fp32 NCDHW, the 8-channel side tensor is 16-bit */
int ctu_lp_conv3d_first_fwd(int dtype, const float* x, int cin, const float* w, const float* bias, int nbias, void* out,
                            int out_cs, int Co, float* stats, int N, int D, int H, int W, const ctu_bn_tail* tail, void* stream);
int ctu_lp_conv3d_first_bwd_data(int dtype, const void* g, int g_cs, const float* w, int cin, int Co, float* dx, int N,
                                 int D, int H, int W, void* stream);
int ctu_lp_conv3d_first_wgrad(int dtype, const float* x, int cin, const void* g, int g_cs, float* dw, int Co, float* ws,
                              int N, int D, int H, int W, void* stream);
/* the same data gradient on the matrix pipe (volumes at least 32 wide, cin <= 4): an 8 -> 8 "pair"-layout launch whose first
 * cin outputs are stored as the float32 planes of dx.  wp = ctu_lp_pack_conv3d_weight(w, Co, Ci = cin, k = 3, rin_p = 8,
 * nout_p = 8, mode 1, layout 1).  78 -> 33 us at 128^3 (cin 1), 0.70 -> 0.29 ms at 256^3 (cin 2), measured. */
/* kernel symbol a 16-bit forward / weight-gradient launch of this geometry runs (measurement tags, as ctu_conv3d_*_kernel_name) */
const char* ctu_lp_conv3d_fwd_kernel_name(int N, int D, int H, int W, int k, int rin_p, int nout_p, int layout);
const char* ctu_lp_conv3d_wgrad_kernel_name(int D, int H, int W, int k, int cin_p, int cout_p);
int ctu_lp_conv3d_first_bwd_data_pair_supported(int cin, int W);
int ctu_lp_conv3d_first_bwd_data_pair(int dtype, const void* g, int g_cs, const void* wp, int cin, float* dx,
                                      int N, int D, int H, int W, void* stream);
/* ConvTranspose3d(C, C, 2, 2) + bias; packing mode 0 forward / 1 data gradient (different sizes: packed_elems(mode)) */
size_t ctu_lp_convt_packed_elems(int rin_p, int nout_p, int mode);
int ctu_lp_pack_convt_weight(int dtype, const float* w, void* wp, int Ci, int Co, const int32_t* cinv, int rin_p,
                             int nout_p, int mode, void* stream);
int ctu_lp_convt2_fwd(int dtype, const void* in, int in_cs, int rin_p, const float* in_scale, const float* in_shift,
                      int in_relu, const void* wp, const float* bias, int nbias, void* out, int out_cs, int nout_p,
                      int N, int D, int H, int W, void* stream);
int ctu_lp_convt2_bwd_data(int dtype, const void* gout, int g_cs, int rout_p, const void* wp, void* gin, int gin_cs,
                           int nin_p, int N, int D, int H, int W, void* stream);
size_t ctu_lp_convt2_wgrad_ws_floats(int N, int D, int H, int W, int cin_p, int cout_p);
/* dw only (fp32 [Ci,Co,2,2,2]); the bias gradient is ctu_lp_channel_sum of gout */
int ctu_lp_convt2_wgrad(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                        int in_relu, const void* gout, int g_cs, int cout_p, float* dw, int Ci, int Co,
                        const int32_t* imap, float* ws, int N, int D, int H, int W, void* stream);
/* HBM-bound glue: same kernels as the fp32 entry points, instantiated for 16-bit storage */
/* 16-bit fused decoder up-convolution (upconv_lp.hip; ConvTranspose3d(C,C,2,2) -> Conv3d(C,Co<=8,3,p=1), models.py:37-38) on
 * v_mfma_f32_16x16x32_{bf16,f16}: the twins of ctu_upconv_fused_fwd / _wgrad / _project / _bwd_data for 16-bit tensors.
 * Supported: 8 padded output channels, input channels a multiple of 32, coarse width >= 16 (the decoder's top level of the
 * shipped nets); other geometries take the unfused ctu_lp_convt2_* + ctu_lp_conv3d_* kernels.  N, D, H, W: COARSE dims.
 *  pack ....... wp32 = ctu_upconv_fused_pack's fp32 packing (cin_p, nout_p = 8) -> wp16 [ctu_lp_upconv_fused_packed_elems]
 *               (forward fragments, then data-gradient fragments), rounded once from the fp32 composite weights
 *  fwd ........ out (fine grid, 8 channels, raw) + stats [ctu_lp_upconv_fused_num_blocks][2][8] of the ROUNDED outputs
 *  wgrad ...... dweff fp32 [8][8][cin_p][8] (gradient channel stride must be 8); then ctu_lp_upconv_fused_project
 *  bwd_data ... gin (coarse, cin_p channels) from the fine-grid raw-output gradient */
int ctu_lp_upconv_fused_supported(int k, int D, int H, int W, int cin_p, int nout_p);
size_t ctu_lp_upconv_fused_packed_elems(int cin_p);
int ctu_lp_upconv_fused_num_blocks(int N, int D, int H, int W);
int ctu_lp_upconv_fused_pack(int dtype, const float* wp32, int cin_p, void* wp16, void* stream);
int ctu_lp_upconv_fused_fwd(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                            int in_relu, const void* wp16, const float* beff, void* out, int out_cs, float* stats,
                            int N, int D, int H, int W, void* stream);
size_t ctu_lp_upconv_fused_wgrad_ws_floats(int N, int D, int H, int W, int cin_p);
int ctu_lp_upconv_fused_wgrad(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                              int in_relu, const void* gout, int g_cs, float* dweff, float* ws, int N, int D, int H,
                              int W, void* stream);
/* ... with the BatchNorm + ReLU backward of the fused op's output folded in (as ctu_lp_conv3d_wgrad_bn): gy_out feeds
 * ctu_lp_upconv_fused_project and ctu_lp_upconv_fused_bwd_data */
int ctu_lp_upconv_fused_wgrad_bn_supported(int N, int D, int H, int W, int cin_p);
int ctu_lp_upconv_fused_wgrad_bn(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                                 int in_relu, const void* ga, int g_cs, const void* y, const float* bn_scale,
                                 const float* bn_shift, const float* coef, void* gy_out, float* dweff, float* ws,
                                 int N, int D, int H, int W, void* stream);
int ctu_lp_upconv_fused_project(int dtype, const float* dweff, const void* gout, int g_cs, int nout_p, int N, int D, int H,
                                int W, const float* bt, const float* pack_ws, const int32_t* imap, int C, int Co, int cin_p,
                                float* dwt, float* dbt, float* dw3, float* ws, void* stream);
int ctu_lp_upconv_fused_bwd_data(int dtype, const void* gout, int g_cs, const void* wp16, void* gin, int gin_cs, int cin_p,
                                 int N, int D, int H, int W, void* stream);
int ctu_lp_ncdhw_to_ndhwc(int dtype, const float* src, void* dst, int N, int C, int D, int H, int W, int cp, int cs,
                          void* stream);
int ctu_lp_ndhwc_to_ncdhw(int dtype, const void* src, float* dst, int N, int C, int D, int H, int W, int cs,
                          void* stream);
int ctu_lp_bn_relu_bwd_reduce(int dtype, const void* y, int y_cs, const void* ga, int g_cs, int cp, const float* scale,
                              const float* shift, const float* mean, const float* invstd, int64_t nvox,
                              float* partials, const ctu_bn_bwd_tail* tail, void* stream);
int ctu_lp_bn_relu_bwd_apply(int dtype, const void* y, int y_cs, void* ga, int g_cs, int cp, const float* scale,
                             const float* shift, const float* mean, const float* invstd, const float* coef,
                             int64_t nvox, void* stream);
int ctu_lp_maxpool2_fwd(int dtype, const void* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                        int in_relu, void* out, int out_cs, int N, int D, int H, int W, void* stream);
int ctu_lp_maxpool2_bwd(int dtype, const void* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                        int in_relu, const void* gout, int gout_cs, void* gin, int gin_cs, int accumulate,
                        int N, int D, int H, int W, void* stream);
int ctu_lp_maxpool2_bwd_bn(int dtype, const void* in, int in_cs, int cp, const float* in_scale, const float* in_shift,
                           const float* mean, const float* invstd, const void* gout, int gout_cs, void* gin,
                           int gin_cs, int accumulate, int N, int D, int H, int W, float* partials, const ctu_bn_bwd_tail* tail, void* stream);
int ctu_lp_skip_add(int dtype, const void* a, int a_cs, const float* a_scale, const float* a_shift, int a_relu,
                    const void* b, int b_cs, const float* b_scale, const float* b_shift, int b_relu,
                    void* out, int out_cs, int cp, int64_t nvox, void* stream);
int ctu_lp_channel_sum(int dtype, const void* x, int cs, int cp, int64_t nvox, float* partials, float* out, int C,
                       void* stream);
/* head: 16-bit input / input gradient, fp32 NCDHW outputs and output gradients (the loss stays fp32).
 * ctu_lp_head_bwd_bn's gscale multiplies the incoming output gradients g0 / g1 as they are read: the float16 loss scale
 * (1 otherwise) -- everything the launch produces is linear in them, so no scaled copy of the two output-sized maps is made. */
int ctu_lp_head_fwd(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                    int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                    int head_mode, float* out0, float* out1, int N, int64_t nvox_per_item, void* stream);
int ctu_lp_head_bwd_bn(int dtype, const void* in, int in_cs, int cin_p, const float* in_scale, const float* in_shift,
                       int in_relu, const float* w, const float* bias, const int32_t* imap, int Ci, int Co, int act,
                       int head_mode, const float* g0, const float* g1, void* gin, int gin_cs, float* dw,
                       float* db, float* ws, int N, int64_t nvox_per_item, const float* bn_mean,
                       const float* bn_invstd, int bn_cp, float* bn_partials, const ctu_bn_bwd_tail* tail, float gscale,
                       void* stream);
/* Every tensor of a list scaled in place by s (one launch): un-scaling of loss-scaled fp16 gradients.
 * ptrs: HOST array of n DEVICE float pointers, sizes: HOST int64[n].  nonfinite_flag: NULL or a DEVICE float[1] that is set
 * to 1 when any scaled value is inf / NaN (the fp16 backward overflowed; the caller zeroes it per step and hands it to
 * ctu_adam_amsgrad as skip_flag, what torch.amp.GradScaler does with found_inf). */
int ctu_scale_tensors(void* const* ptrs, const int64_t* sizes, int n, float s, float* nonfinite_flag, void* stream);

/* --------------------------------------------------------------- gradient exchange (RCCL over xGMI) ---- */
/* One process per GPU; the only cross-GPU step of the path is the mean of the parameter gradients before the optimizer
 * step.  Replaces nn.DataParallel's replicate / scatter / gather / reduce-add (ctunet/pytorch/Model.py:481-487).
 *   ctu_comm_available ... 1 if librccl could be bound (it is dlopen'ed by soname on first use; no link-time dependency)
 *   ctu_comm_unique_id ... HOST buffer of CTU_COMM_ID_BYTES, filled on ONE rank and handed to the others by the caller
 *                          (any side channel: a torch.distributed store / broadcast, MPI, a file)
 *   ctu_comm_init ........ collective over all ranks (ncclCommInitRank) on the calling thread's current HIP device
 *   ctu_comm_allreduce_f32 ncclAllReduce(sum | avg) of `count` floats on the caller's stream; send == recv allowed;
 *                          returns without synchronising.  Not capturable: launch it between graph segments.
 *   ctu_comm_destroy ..... ncclCommDestroy */
#define CTU_COMM_ID_BYTES 128
int ctu_comm_available(void);
int ctu_comm_unique_id(void* id_host);
int ctu_comm_init(void** comm, int world, int rank, const void* id_host);
int ctu_comm_allreduce_f32(void* comm, const float* send, float* recv, size_t count, int average, void* stream);
int ctu_comm_destroy(void* comm);

/* Per-channel sum over voxels of a channels-last tensor: out[c] = sum_v x[v,c] (bias grads). */
int ctu_channel_sum_num_blocks(int64_t nvox);
int ctu_channel_sum(const float* x, int cs, int cp, int64_t nvox, float* partials,
                    float* out, int C, void* stream);
/* Fused multi-tensor Adam / AdamW with amsgrad (torch.optim.Adam(amsgrad=True) and optim.AdamW,
 * Model.py:514-527).  ptrs: HOST array of 5*n DEVICE pointers {param, grad, exp_avg, exp_avg_sq,
 * max_exp_avg_sq} (copied into the kernel arguments, 64 tensors per launch); sizes: HOST int64[n].
 * step: DEVICE float[1] step counter, incremented by this call before it is used (so a captured graph
 * keeps advancing the bias corrections).  decoupled != 0: AdamW weight decay.  skip_flag: NULL or a DEVICE float[1];
 * non-zero = skip this update entirely (parameters, moments and the step counter untouched): an overflowed fp16 step. */
int ctu_adam_amsgrad(void* const* ptrs, const int64_t* sizes, int n, float* step,
                     double lr, double beta1, double beta2, double eps, double weight_decay,
                     int decoupled, const float* skip_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTUNET_HIP_H */
