/* eat_hip.h -- C ABI of libeat_hip.so: the MI355X (gfx950) hot path of EfficientAT.
 *
 * The reference (fschmid56/EfficientAT) is pure Python on PyTorch: its "FFI" for this path
 * is the set of torch ops its modules call.  Each entry point below replaces the torch-op
 * sequence of one reference call site (cited per function, paths relative to the reference
 * root).  A maintainer binds them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name starts with h_; fp32 unless noted
 *   - tensors are contiguous NCHW = (B, C, F, T), T fastest; S = F*T is the plane size
 *   - the caller owns every buffer; the library allocates nothing and keeps no mutable
 *     global state - with ONE documented exception, the process-wide kernel-variant selector
 *     eat_pw_stream_mode() (an atomic int read once per eat_pw_conv_bf16_fwd call; it selects between
 *     kernels with identical results) - so calls may be issued concurrently on different streams
 *   - environment variables read by the library (each once, at first use): EAT_PW_STREAM (initial
 *     value of that selector) and EAT_WGRAD_FP32 (=1: every 1x1 weight gradient on the exact fp32
 *     kernel, a debugging override of the caller's `exact_fp32` argument); nothing else
 *   - work is enqueued on `stream` (a hipStream_t) asynchronously; no hidden synchronisation
 *   - return 0 on success, a negative EAT_E* code otherwise (never throws, never exits);
 *     eat_last_error_string() returns the message of the calling thread's last failure
 */
#ifndef EAT_HIP_H
#define EAT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#pragma GCC visibility push(default)

typedef void* eat_stream_t; /* hipStream_t */

#define EAT_OK 0
#define EAT_EINVAL (-1)   /* unsupported shape / bad argument            */
#define EAT_ELAUNCH (-2)  /* HIP reported an error at kernel launch      */

/* activation codes shared by all conv entry points */
#define EAT_ACT_NONE 0
#define EAT_ACT_RELU 1
#define EAT_ACT_HSWISH 2   /* x * clamp(x + 3, 0, 6) / 6 (nn.Hardswish) */
#define EAT_ACT_SIGMOID 3  /* eat_linear_fwd only (SqueezeExcitation gate)  */

int eat_version(void);
const char* eat_last_error_string(void);

/* ---- mel front-end: models/preprocess.py:40-67 (AugmentMelSTFT.forward) -----------------
 * wave (B, L) -> out (B, n_mels, T), T = 1 + (L - 1) / hop  [centre-padded STFT of the
 * pre-emphasised signal, which has L-1 samples].
 * One fused kernel: pre-emphasis conv1d [-0.97, 1] (:41), reflect pad n_fft/2 + hann window
 * + rFFT (:42-43), power (:44), mel filterbank matmul (:56-57), log(x + 1e-5) (:59),
 * optional frequency / time masking (:61-63), (x + 4.5) / 5 (:65).
 *   window      (win_length)   the reference's torch.hann_window(win_length, periodic=False)
 *   twiddle     (n_fft, 2)     exp(-2*pi*i*j/n_fft) as (cos, -sin) pairs, fp32 from fp64
 *   band_w2     (band_pairs, n_mels, 2)  the non-zero band of each row of the kaldi mel basis (:52-55), built on
 *                              the host with the reference's fp32 op order, as 8-byte pairs: pair j of row m =
 *                              basis[m][band_start[m] + 2j .. +1]; pairs beyond band_cnt[m] are zero
 *   band_start  (n_mels)       EVEN first FFT bin of each band; band_start[m] + 2*band_pairs <= n_fft/2
 *   band_cnt    (n_mels)       pairs that cover row m's non-zeros (the kernel walks max(band_cnt) per 64 rows)
 *   mask_f0..mask_t1           [f0,f1) mel rows and [t0,t1) frames set to 0.0 after the log
 *                              (train-mode masking); pass f0==f1 / t0==t1 for none
 * Only n_fft == 1024 and n_mels <= 256 are implemented (every reference config); others return EAT_EINVAL. */
int eat_mel_fwd(const float* wave, int B, int L, const float* window, int win_length, int n_fft,
                int hop, const float* twiddle, const float* band_w2, const int* band_start,
                const int* band_cnt, int n_mels, int band_pairs, float* out, int T, int mask_f0,
                int mask_f1, int mask_t0, int mask_t1, eat_stream_t stream);

/* ---- stem: models/mn/model.py:124-133 (ConvNormActivation 3x3 stride 2, 1 -> C) ----------
 * x (B,1,F,T) -> y (B,C,Fo,To), Fo=(F+1)/2-ish per cnn_out_size; w (C,1,3,3) with the eval-mode
 * BatchNorm already folded in (w' = w*g/sqrt(rv+eps)), bias (C) = b - rm*g/sqrt(rv+eps). */
int eat_stem_conv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C,
                      int F, int T, int Fo, int To, int act, eat_stream_t stream);

/* ---- depthwise k x k conv: models/mn/block_types.py:150-162 ------------------------------
 * x (B,C,F,T) -> y (B,C,Fo,To); k in {3,5}, stride in {1,2}, pad (k-1)/2; w (C,k,k) BN-folded,
 * bias (C).  If pool != NULL, the per-(b,c) sum over the output plane of the ACTIVATED output
 * is atomically accumulated into pool (B,C) (zeroed by the caller): the squeeze of
 * SqueezeExcitation (block_types.py:72-73) fused into the producer. */
int eat_dw_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* pool,
                    int B, int C, int F, int T, int Fo, int To, int k, int stride, int act,
                    eat_stream_t stream);

/* The same with a dilation (models/mn/model.py:244-269 `dilated=True`; block_types.py:150: stride forced to 1, padding
 * (k-1)/2*dilation): generic thread-per-output kernel, any odd k <= 7 / stride / dilation. */
int eat_dw_conv_dilated_fwd(const float* x, const float* w, const float* bias, float* y, float* pool, int B, int C,
                            int F, int T, int Fo, int To, int k, int stride, int dilation, int act, eat_stream_t stream);

/* Backward of the dilated depthwise conv (training of the `dilated=True` networks, models/mn/model.py:244-269: autograd
 * of F.conv2d(..., dilation) in block_types.py:150-162): dx (B,C,F,T) from dz (B,C,Fo,To) and the taps w (C,k*k);
 * dw (C,k*k) = sum over batch and positions of dz times the shifted input (plain stores: one block per tap and channel). */
int eat_dw_conv_dilated_dgrad(const float* dz, const float* w, float* dx, int B, int C, int F, int T, int Fo, int To,
                              int k, int stride, int dilation, eat_stream_t stream);
int eat_dw_conv_dilated_wgrad(const float* dz, const float* x, float* dw, int B, int C, int F, int T, int Fo, int To,
                              int k, int stride, int dilation, eat_stream_t stream);

/* ---- pointwise 1x1 conv (GEMM): models/mn/block_types.py:138-147,167-171; mn/model.py:159-167
 * y[b] (Co,S) = act( W (Co,Ci) . (x[b] (Ci,S) * in_scale[b,:,None]) + bias[:,None] ) + res[b]
 * wp = W packed by eat_pw_prepack (BN scale folded in via row_scale), bias (Co);
 * in_scale (B,Ci) or NULL (SE scale, block_types.py:83);
 * res (B,Co,S) or NULL (block_types.py:179-180, added after the activation-free project conv);
 * pool (B,Co) or NULL: per-(b,co) sums of the final output (global average pool of
 * mn/model.py:220 fused; y may then be NULL to skip materialising the last feature map).
 * fp32 MFMA (v_mfma_f32_16x16x4_f32): exact fp32 products and accumulation.
 * Requires Ci % 4 == 0 and S % 4 == 0 (true for every reference configuration). */
int eat_pw_conv_fwd(const float* x, const float* wp, const float* bias, const float* in_scale,
                    const float* res, float* y, float* pool, int B, int Ci, int Co, int S, int act,
                    eat_stream_t stream);

/* Pack w (Co,Ci) * row_scale[:,None] (row_scale may be NULL) into MFMA A-fragment order:
 * wp has (Ci/4) * ceil(Co/16) * 64 floats.  Done once per weight update, not per step. */
int eat_pw_prepack(const float* w, const float* row_scale, float* wp, int Co, int Ci,
                   eat_stream_t stream);
/* The same pack of the TRANSPOSE of a stored matrix: w_t is (Ci, Co) row-major, the packed matrix is w_t^T (Co, Ci) -
 * the data-gradient GEMM dx = W^T dz of the 1x1 convs (autograd of block_types.py:138-147,167-171) without a transposed
 * copy of the weights. */
int eat_pw_prepack_t(const float* w_t, const float* row_scale, float* wp, int Co, int Ci, eat_stream_t stream);

/* ---- linear: models/mn/model.py:189-193 (classifier Linear layers) and the two Linear layers
 * of SqueezeExcitation (models/mn/block_types.py:64-65,74-79: fc1+ReLU, fc2+Sigmoid) ---------
 * y (B,N) = act( (x (B,K) * x_scale) . W^T (N,K) + bias ); x_scale multiplies x (1/S turns
 * pooled sums into the mean of mn/model.py:220). */
int eat_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                   float x_scale, int act, eat_stream_t stream);

/* ================= training step (batch-statistics BatchNorm + backward) ======================
 * The reference trains with nn.BatchNorm2d(eps=1e-3, momentum=0.01) in train mode and leaves the
 * backward pass to autograd (ex_audioset.py:147-199).  Formulas: SURVEY.md Appendix C. */

/* Batch statistics of nn.BatchNorm2d in train mode (norm layer of every ConvNormActivation:
 * models/mn/model.py:114-115, block_types.py:138-171).
 * sums (2C doubles, zeroed by the caller): sums[c] += sum z, sums[C+c] += sum z^2 over (B, S). */
int eat_bn_stats(const float* z, int B, int C, int S, double* sums, eat_stream_t stream);

/* From the sums: batch mean, biased variance -> a = gamma*invstd, b = beta - mean*a (so that
 * BN(z) = a*z + b), saved mean / invstd for backward; if running_mean != NULL the running
 * buffers get the momentum update with the unbiased variance (n = B*S elements per channel). */
int eat_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, double n, int C, float* a, float* b,
                    float* mean, float* invstd, eat_stream_t stream);

/* y = act(a_c*z + b_c) [+ res]; pool (B,C) or NULL receives the plane sums of y (SE squeeze,
 * block_types.py:72-73, or the head's global average pool); y may be NULL when only pool is needed. */
int eat_bn_act_fwd(const float* z, const float* a, const float* b, const float* res, float* y,
                   float* pool, int B, int C, int S, int act, eat_stream_t stream);

/* autograd of BatchNorm2d (train) + activation [+ the SE multiply of block_types.py:83] in
 * ex_audioset.py:197 (loss.backward()):
 * Backward of y = act(a*z + b) under batch statistics, with the incoming gradient
 * g_in[b,c,s] = dy[b,c,s]*gscale[b,c] + gadd[b,c] (gscale/gadd may be NULL: the SE scale and the
 * broadcast gradient of the squeeze are folded in here).  Pass 1 accumulates per channel
 * sums[c] += sum g, sums[C+c] += sum g*xhat with g = g_in*act'(a z + b) (= dbeta, dgamma);
 * pass 2 writes dz = a*(g - sums[c]/n - xhat*sums[C+c]/n). */
int eat_bn_act_bwd_reduce(const float* dy, const float* z, const float* a, const float* b,
                          const float* mean, const float* invstd, const float* gscale,
                          const float* gadd, int B, int C, int S, int act, double* sums,
                          eat_stream_t stream);
int eat_bn_act_bwd_apply(const float* dy, const float* z, const float* a, const float* b,
                         const float* mean, const float* invstd, const float* gscale,
                         const float* gadd, const double* sums, float* dz, int B, int C, int S,
                         int act, eat_stream_t stream);

/* autograd of the SE scale multiply (models/mn/block_types.py:83):
 * out[b,c] = sum_s u[b,c,s] * v'[b,c,s] with v' = v (a == NULL) or act(a_c*v + b_c): the gradient
 * w.r.t. the SE scale, d s[b,c] = sum_s d(x*s) * x. */
int eat_plane_dot(const float* u, const float* v, const float* a, const float* b, float* out, int B,
                  int C, int S, int act, eat_stream_t stream);

/* ---- round 3: BatchNorm statistics / backward reductions without their own passes -------------------------
 * Reference semantics throughout: nn.BatchNorm2d(eps=1e-3, momentum=0.01) in train mode (models/mn/model.py:114-115)
 * around the convs of InvertedResidual (models/mn/block_types.py:138-171); backward = autograd of those modules
 * (ex_audioset.py:190-199).  Partial-sum buffers are plain fp32 arrays written with ordinary stores (no atomics, no
 * zero fill) by the producing kernel and reduced per channel in fp64 by the finalize kernels. */

/* Upper bound of the partial slots per (b,c) plane that eat_dw_conv_fwd_stats (dgrad = 0) / eat_dw_conv_dgrad_g
 * (dgrad = 1) write for this geometry: size the buffers with it. Host-only helper (no device work). */
int eat_dw_partials_inner(int F, int T, int Fo, int To, int k, int stride, int dgrad);

/* Train-mode depthwise conv of models/mn/block_types.py:150-162: y = conv(act_in(in_a[c] x + in_b[c])) (in_a == NULL:
 * plain x; the transform is the BatchNorm + activation of the expand conv evaluated on load) PLUS the batch statistics
 * of y for the BatchNorm that follows: part [B][2][C][inner] floats (sum, sum of squares per wave);
 * *h_inner (host int) receives inner <= inner_cap. */
int eat_dw_conv_fwd_stats(const float* x, const float* in_a, const float* in_b, int in_act, const float* w, float* y,
                          float* part, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo, int To, int k,
                          int stride, eat_stream_t stream);

/* The same partials from a tensor that already exists (any geometry; inner = 1): part [B][2][C]. */
int eat_bn_stats_partial(const float* z, int B, int C, int S, float* part, eat_stream_t stream);

/* (a, b, mean, invstd) and the running-buffer update of nn.BatchNorm2d from partials [outer][2][C][inner]
 * (n = elements per channel). Same outputs as eat_bn_finalize.  ws: NULL, or eat_bn_finalize_ws_doubles(outer, C, inner)
 * doubles of scratch (> 0 when there are many partial rows: the rows are then summed in groups with coalesced reads
 * first - two launches, fixed summation order). */
int eat_bn_finalize_ws_doubles(int outer, int C, int inner);
int eat_bn_finalize_partials(const float* part, int outer, int C, int inner, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, double n, float* a,
                             float* b, float* mean, float* invstd, double* ws, eat_stream_t stream);

/* Data-gradient GEMM of the project conv, y = W^T dz_p (wp: eat_pw_prepack_t / eat_pw_prepack_bf16_t pack of the project
 * weight, wmode 0 = fp32, 2 = bf16 hi / lo), with the backward statistics of the depthwise conv's BatchNorm + activation
 * (autograd through block_types.py:150-162) in its epilogue: part [tiles][2][Co], tiles = eat_pw_conv_stat_tiles(B, S, 0),
 * holds per 256-column tile the sums of g and of g (z_d - c), c = -g_b / g_a, with g = y * act'(g_a z_d + g_b); gz = z_d (B, Co, S).  Replaces
 * eat_bn_act_bwd_reduce over (y, z_d) where no SE gate sits between the two (gscale / gadd = NULL there).  Returns 1 and
 * launches nothing when S % 4 != 0.  eat_bn_bwd_sums_from_tiles turns the partials into the sums eat_bn_act_bwd_apply /
 * eat_dw_conv_bwd_bn_g read (ws: eat_bn_bwd_sums_ws_doubles(tiles, C) doubles, NULL when that is 0). */
int eat_pw_conv_gstats_fwd(const float* x, const void* wp, int wmode, const float* zero_bias, float* y, const float* gz,
                           const float* g_a, const float* g_b, int g_act, float* part, int B, int Ci, int Co, int S,
                           eat_stream_t stream);
int eat_bn_bwd_sums_ws_doubles(int tiles, int C);
/* g_a / g_b: the BatchNorm's (a, b) the partials were taken with - the epilogue accumulates sum g (z_d - c), c = -g_b / g_a (the
 * zero of the pre-activation: no cancellation against a large channel mean in the fp32 tile partials), and this call adds
 * (c - mean) sum g back in fp64; both NULL: partials of the uncentred sum g z_d. */
int eat_bn_bwd_sums_from_tiles(const float* part, int tiles, int C, const float* mean, const float* invstd, const float* g_a,
                               const float* g_b, double* ws, double* sums, eat_stream_t stream);

/* Centred Gram matrix of a conv input x (B, C, S) (the statistics of the conv1x1 -> nn.BatchNorm2d pair of
 * models/mn/block_types.py:138-147 without reading the conv output): Gc = sum_{b,s} (x - m)(x - m)^T, m = sx * inv_n the
 * channel means - both operands of the weight-gradient kernel are centred on load.  w^T Gc w = sum (z - w.m)^2 is n var(z)
 * DIRECTLY: with the plain Gram matrix the variance is the difference of two sums (mu/sigma)^2 times larger (a channel
 * with |mean| = 30 std loses 3 digits), and an error of the mean enters only in second order.  ws / n_slots as
 * eat_pw_conv_wgrad_ws (n_slots >= eat_pw_wgrad_slots(B, C, C, S, exact, 1): bit-reproducible). */
int eat_gram_centered(const float* x, const float* sx, float inv_n, float* G, float* ws, int n_slots, int B, int C, int S,
                      int exact_fp32, eat_stream_t stream);     /* ws: 2*C + n_slots*C*C floats, zero-filled */


/* BatchNorm state of the expand conv z = W x (models/mn/block_types.py:138-147) from the Gram matrix of its input:
 * Tm = W G (Co x Ci) with G = sum_{b,s} x x^T, sx = sum_{b,s} x (Ci): sum z = W sx, sum z^2 = rowsum(Tm .* W).
 * centered != 0: G is eat_gram_centered's Gc: var = rowsum(Tm .* W) / n.
 * The (3-6x wider) tensor z is not read for its statistics. Outputs as eat_bn_finalize. */
int eat_gram_bn_finalize(const float* Tm, const float* W, const float* sx, int Co, int Ci, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         double n, float* a, float* b, float* mean, float* invstd, int centered, eat_stream_t stream);

/* The same from G itself (Ci x Ci): Tm = W G (Co x Ci) is formed in fp64 inside and written out (the backward's
 * eat_expand_bwd_coef reads it) - no separate GEMM launch for W G. */
int eat_gram_bn_finalize_g(const float* G, const float* W, const float* sx, int Co, int Ci, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, float momentum, float eps, double n,
                           float* Tm, float* a, float* b, float* mean, float* invstd, int centered, eat_stream_t stream);

/* g = dy * act'(a[c] z + b[c]) (g may alias dy) and gpart[b*C + c] = sum_s g: the first half of the backward of
 * act(BatchNorm(z)) as a stand-alone pass (geometries where eat_dw_conv_dgrad_g has no fused kernel). */
int eat_act_grad_sum(const float* dy, const float* z, const float* a, const float* b, int act, float* g, float* gpart,
                     int B, int C, int S, eat_stream_t stream);

/* Squeeze-excitation blocks (models/mn/block_types.py:72-83 between the depthwise BN + act and the project conv): one
 * pass over d = d(x * s) and the pre-BN z for BOTH the gate gradient and the BatchNorm-backward sums.  P (5, B*C):
 * P0 = sum_s d y (= d s, what eat_plane_dot(d, z, a, b) returns), P1 = sum d act', P2 = sum act',
 * P3 = sum d act' (z - mean), P4 = sum act' (z - mean), with y = act(a z + b), act' at a z + b. */
int eat_se_bn_bwd_partials(const float* d, const float* z, const float* a, const float* b, const float* mean, float* P,
                           int B, int C, int S, int act, eat_stream_t stream);

/* ... and, once the gate MLP's backward has produced gadd (B,C) (gscale = the gate s): the per-channel sums that
 * eat_bn_act_bwd_apply takes, sums[c] = sum_b (s P1 + gadd P2), sums[C+c] = invstd[c] sum_b (s P3 + gadd P4) -
 * i.e. eat_bn_act_bwd_reduce(d, z, ..., gscale, gadd) without reading d and z again. */
int eat_se_bn_bwd_combine(const float* P, const float* gscale, const float* gadd, const float* invstd, int B, int C,
                          double* sums, eat_stream_t stream);

/* Depthwise data gradient (autograd of block_types.py:150-162) whose epilogue starts the backward of the expand conv's
 * BatchNorm + activation: g = dgrad(dz) * act'(ga[c] gz + gb[c]), gz = pre-BN output of the expand conv (B,C,F,T);
 * gpart [B][C][inner] = per-wave sums of g; *h_inner receives inner <= inner_cap. */
int eat_dw_conv_dgrad_g(const float* dz, const float* w, const float* gz, const float* ga, const float* gb, int gact,
                        float* g, float* gpart, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo, int To,
                        int k, int stride, eat_stream_t stream);

/* Partial slots per plane of eat_dw_conv_bwd_g's merged kernel (host helper). */
int eat_dw_bwd_partials_inner(int F, int T, int Fo, int To, int k, int stride);

/* Backward of the depthwise conv (autograd of models/mn/block_types.py:150-162) in ONE pass over dz (B,C,Fo,To) and the
 * pre-BN expand output x (B,C,F,T): dw (C,k,k) += weight gradient w.r.t. act(in_a x + in_b) [dw zeroed by the caller],
 * g = dgrad(dz) * act'(in_a x + in_b) and its per-tile sums gpart [B][C][inner] - i.e. eat_dw_conv_wgrad_tf +
 * eat_dw_conv_dgrad_g with every tensor read once.  inner_cap >= max(eat_dw_bwd_partials_inner, eat_dw_partials_inner(.., 1)). */
int eat_dw_conv_bwd_g(const float* dz, const float* x, const float* in_a, const float* in_b, int in_act, const float* w,
                      float* g, float* dw, float* gpart, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo,
                      int To, int k, int stride, eat_stream_t stream);

/* Backward of conv1x1 (W: Co x Ci) -> BatchNorm(train) -> act WITHOUT forming dz (autograd of block_types.py:138-147):
 * with g as above, Gx = sum g x^T (eat_pw_conv_wgrad(g, x)), Tm = W G, sx, the forward's (a, mean, invstd):
 *   dgamma = invstd (rowsum(W .* Gx) - mean S1), dbeta = S1 = sum gpart, m1 = S1/n, m2 = dgamma/n (0 when frozen)
 *   dW  = diag(a) [Gx - m1 sx^T - diag(m2 invstd)(Tm - mean sx^T)]
 * and the operands of dx = WaT g + M x + c0 (two eat_pw_conv_fwd launches), all Ci x Co transposes so that
 * eat_linear_fwd (which contracts over the contiguous axis) forms M = W2T . WT^T (Ci x Ci) and c0 = e1 . WT^T (Ci):
 *   WaT = (diag(a) W)^T,  WT = W^T,  W2T = -(diag(a m2 invstd) W)^T,  e1 = a (m2 invstd mean - m1) (Co).
 * centered != 0: Tm = W Gc (eat_gram_centered), i.e. "Tm - mean sx^T" is already inside the operand.
 * WaT / WT / W2T may be NULL together (round 6): the caller then takes e2 = a m2 invstd (Co; may be NULL otherwise) and
 * forms the operands with eat_expand_bwd_wcat. */
int eat_expand_bwd_coef(const float* W, const float* Gx, const float* Tm, const float* sx, const float* gpart, int outer,
                        int inner, int Co, int Ci, const float* a, const float* mean, const float* invstd, double n,
                        int frozen, float* dW, float* dgamma, float* dbeta, float* WaT, float* WT, float* W2T, float* e1,
                        int centered, float* e2, eat_stream_t stream);

/* The operands of dx = [WaT | M] [g ; x] + c0 (previous comment) written straight as the weight pack of the two-source
 * data-gradient GEMM (eat_pw_conv_cat_fwd / eat_pw_conv_b16_fwd with x2) - one launch instead of transposes + pack + GEMM +
 * concatenation + pack:  Wcat (Ci x (Co + Ci)) = [ (diag(a) W)^T | -W^T diag(e2) W ],  c0 = W^T e1 (Ci).
 * kind 0: fp32 fragments (the layout of eat_pw_prepack), 1: bf16, 2: bf16 hi + lo (eat_pw_prepack_bf16); wp holds
 * eat_expand_bwd_wcat_elems(Co, Ci, kind) elements; M is accumulated on the exact fp32 MFMA whatever the kind. */
int eat_expand_bwd_wcat_elems(int Co, int Ci, int kind);
int eat_expand_bwd_wcat(const float* W, const float* a, const float* e2, const float* e1, int Co, int Ci, int kind, void* wp,
                        float* c0, eat_stream_t stream);

/* eat_dw_conv_bwd_g with the BatchNorm + activation backward of the depthwise conv's OWN output evaluated on load
 * (autograd through models/mn/block_types.py:150-162 -> :72-83): dy (B,C,Fo,To) is the gradient w.r.t. act(BN(z)) - for a
 * squeeze-excitation block the incoming gradient is dy * gscale[b,c] + gadd[b,c] -, z the conv output, bn_* the forward's
 * (a, b, mean, invstd), sums (2C doubles) the channel sums of eat_bn_act_bwd_reduce / eat_se_bn_bwd_combine; frozen != 0:
 * running statistics were used (no batch-mean terms).  dz is never written.  gpart may be NULL.  Only where eat_dw_bwd_merged_ok(...) != 0 (host helper), else EAT_EINVAL. */
int eat_dw_bwd_merged_ok(int B, int C, int F, int T, int Fo, int To, int k, int stride);
int eat_dw_conv_bwd_bn_g(const float* dy, const float* z, const float* bn_a, const float* bn_b, const float* bn_mean,
                         const float* bn_invstd, const float* gscale, const float* gadd, const double* sums, int bn_act,
                         int frozen, const float* x, const float* in_a, const float* in_b, int in_act, const float* w,
                         float* g, float* dw, float* gpart, int inner_cap, int* h_inner, int B, int C,
                         int F, int T, int Fo, int To, int k, int stride, eat_stream_t stream);

/* Backward of the squeeze-excitation gate MLP (autograd through models/mn/block_types.py:72-83:
 * scale = sigmoid(fc2(relu(fc1(mean y))))) in two launches, operands addressed by strides (no transposed copies):
 *   in : ds (B,C) = sum_{f,t} d_out * y, scale (B,C), h (B,Cr) = relu(fc1(..)), pool (B,C) = sum_{f,t} y, W1 (Cr,C), W2 (C,Cr),
 *        inv_s = 1 / (F T)
 *   out: dW1 (Cr,C), db1 (Cr), dW2 (C,Cr), db2 (C), gadd (B,C) = the gradient w.r.t. y through the squeeze;
 *        dh: scratch of eat_se_mlp_dh_floats(B, C, Cr) floats.  fp32 products and accumulation on the matrix cores, every
 *        output element written by one block or summed from k slices in a fixed order (bit-reproducible). */
int eat_se_mlp_dh_floats(int B, int C, int Cr);
int eat_se_mlp_bwd(const float* ds, const float* scale, const float* h, const float* pool, const float* W1, const float* W2,
                   float inv_s, float* dW1, float* db1, float* dW2, float* db2, float* dh, float* gadd, int B, int C, int Cr,
                   eat_stream_t stream);

/* All weight packs of a pass in one launch: `table` = n device-resident 32-byte records
 * { const float* w; void* wp; int32 Co, Ci, kind, trans; } with kind 0 = eat_pw_prepack, 1 / 2 = eat_pw_prepack_bf16
 * (plain / split) and trans as in the *_t entry points (no row_scale); the packs are bit-identical to those of the
 * single-matrix entry points.  max_threads = max over the records of (Ci / 4) * ceil(Co / 16) * 64 (kind 0) or
 * ceil(Ci / 32) * ceil(Co / 16) * 64 (kinds 1, 2). */
int eat_pw_prepack_multi(const void* table, int n, int max_threads, eat_stream_t stream);

/* Stem of the training step without its pre-activation tensor (models/mn/model.py:124-133 in train mode: Conv2d(1, C, 3,
 * stride 2, padding 1) -> BatchNorm2d -> Hardswish; csrc/stem_train.hip).  The conv is linear in the 9-tap patch p of the
 * log-mel x (B,1,F,T), so its batch statistics follow from G9 = sum p p^T and sp = sum p:
 * eat_stem_gram writes Tm = W G9 (C x 9) and sp (9) (feed them to eat_gram_bn_finalize with Ci = 9), using
 * part (eat_stem_gram_blocks(B, Fo) x 54 floats) as scratch; reduction order fixed (bit-reproducible). */
int eat_stem_gram_blocks(int B, int Fo);
int eat_stem_gram(const float* x, const float* W, float* part, float* Tm, float* sp, int B, int C, int F, int T,
                  eat_stream_t stream);
/* Backward of the same: with g = (dy + dy2) * act'(a[c] (W_c . p) + b[c]) recomputed from the log-mel window,
 * gx (C x 9) = sum g p^T and s1 (C) = sum g; eat_expand_bwd_coef(Ci = 9, inner = outer = 1) turns them into dW, dgamma,
 * dbeta.  dy (B,C,Fo,To); dy2 (same shape) or NULL is added on load (the gradient of the first block's residual branch:
 * its add never needs a pass of its own).  part: scratch of eat_stem_bwd_blocks(B, Fo) x C x 10 floats (per-block sums,
 * added in a fixed order: no atomics). */
int eat_stem_bwd_blocks(int B, int Fo);
int eat_stem_bwd(const float* dy, const float* dy2, const float* x, const float* W, const float* a, const float* b, int act,
                 float* part, float* gx, float* s1, int B, int C, int F, int T, eat_stream_t stream);

/* autograd of the depthwise Conv2d of models/mn/block_types.py:150-162 (data gradient):
 * Depthwise conv data gradient dx (B,C,F,T) from dz (B,C,Fo,To), taps w (C,k,k); res (B,C,F,T) or
 * NULL is added (residual branch gradient). */
int eat_dw_conv_dgrad(const float* dz, const float* w, const float* res, float* dx, int B, int C,
                      int F, int T, int Fo, int To, int k, int stride, eat_stream_t stream);

/* autograd of the depthwise Conv2d (models/mn/block_types.py:150-162) and of the stem Conv2d (mn/model.py:124-133):
 * Depthwise / stem weight gradient dw (C,k,k) += sum dz * shifted x; x has XC = C (depthwise) or
 * XC = 1 (3x3 stem, stride 2) channels.  dw must be zeroed by the caller. */
int eat_dw_conv_wgrad(const float* dz, const float* x, float* dw, int B, int C, int XC, int F, int T,
                      int Fo, int To, int k, int stride, eat_stream_t stream);

/* autograd of the 1x1 Conv2d layers (models/mn/block_types.py:138-147,167-171; mn/model.py:159-167):
 * Pointwise weight gradient dW (Co,Ci) += sum_{b,s} dz[b,co,s] * x[b,ci,s] * x_scale[b,ci]
 * (exact_fp32 = 0: split-operand bf16 MFMA with fp32-class accuracy; exact_fp32 = 1, or a plane size that is not a
 * multiple of 4: exact fp32 MFMA; exact_fp32 = 2: plain bf16 operands with fp32 accumulation - the arithmetic autocast
 * gives the conv weight gradient in the reference's bf16 training; x_scale (B,Ci) or NULL is the SE scale the forward applied to x); dW must be
 * zeroed by the caller.  The data gradient is eat_pw_conv_fwd with the packed W^T. */
int eat_pw_conv_wgrad(const float* dz, const float* x, const float* x_scale, float* dW, int B, int Co,
                      int Ci, int S, int exact_fp32, eat_stream_t stream);
/* eat_pw_conv_wgrad with a workspace ws (n_slots * Co * Ci floats): the streaming kernel of the small matrices (Co, Ci <= 64
 * with one side <= 16, or the Gram matrix dz == x) spreads its atomics over n_slots copies of dW (ws zero-filled) and a
 * second kernel adds them into dW.  With n_slots >= eat_pw_wgrad_slots(...) every block has its own copy: the result is
 * bit-reproducible, and the late-layer shapes (eat_pw_wgrad_kernel_kind == 3) run the wide-tile producer / consumer
 * kernel, which STORES its copies (ws need not be zeroed for it) - without the workspace those shapes run the 128 x 128-tile
 * kernel with atomics into dW. */
int eat_pw_conv_wgrad_ws(const float* dz, const float* x, const float* x_scale, float* dW, float* ws, int n_slots, int B,
                         int Co, int Ci, int S, int exact_fp32, eat_stream_t stream);
/* Copies of dW that give every block of eat_pw_conv_wgrad_ws its own slot: the result is then bit-reproducible from
 * run to run (the copies are added in a fixed order).  same != 0: dz and x are the same tensor (the Gram matrix of
 * eat_gram_bn_finalize, whose round-off reaches the BatchNorm statistics).  Host helper. */
int eat_pw_wgrad_slots(int B, int Co, int Ci, int S, int exact_fp32, int same);
/* Host helper (no launch): the kernel family eat_pw_conv_wgrad / _ws / _tf uses for a shape - 1 = 128 x 128-tile LDS-staged
 * kernel, 2 = exact fp32 kernel, 3 = wide-tile producer / consumer kernel, 10 * (1000 * mtb + 10 * ntb + gram) = the
 * LDS-free streaming kernel with mtb x ntb row tiles per block.  For measurement tools (bench.py's per-kernel byte models). */
int eat_pw_wgrad_kernel_kind(int B, int Co, int Ci, int S, int exact_fp32, int same, int has_scale, int has_tf);

/* Train-mode project conv, models/mn/block_types.py:167-171 fed by :150-162 (+ the SE scale of :72-83): the conv input is
 * act_in(tf_a[k] x + tf_b[k]) * in_scale[b,k] evaluated on the way to the matrix cores, i.e. BatchNorm + activation of
 * the depthwise output z_d are fused into the consumer and the activated tensor is never written.
 * wp: eat_pw_prepack (wmode 0) / eat_pw_prepack_bf16 plain (1) / split (2).  Needs Ci % 8 == 0, S % 4 == 0. */
int eat_pw_conv_tf_fwd(const float* x, const float* tf_a, const float* tf_b, int tf_act, const void* wp, int wmode,
                       const float* bias, const float* in_scale, const float* res, float* y, int B, int Ci, int Co, int S,
                       int act, eat_stream_t stream);

/* 1x1 conv over the channels of TWO tensors x1 (B,C1,S), x2 (B,C2,S): y = W [x1 ; x2] + bias (+ res), W (Co, C1+C2) packed
 * by eat_pw_prepack / eat_pw_prepack_bf16 (wmode as in eat_pw_conv_tf_fwd).  Train plan: autograd of the expand conv
 * (models/mn/block_types.py:138-147) with its BatchNorm correction, dx = [WaT | M] [g ; x] + c0, as one GEMM. */
int eat_pw_conv_cat_fwd(const float* x1, int C1, const float* x2, int C2, const void* wp, int wmode, const float* bias,
                        const float* res, float* y, int B, int Co, int S, int act, eat_stream_t stream);

/* Train-mode 1x1 conv z = W x (no bias / activation / residual) WITH the batch statistics of z for the BatchNorm that
 * follows (models/mn/block_types.py:167-171,177-181; models/dymn/dy_block.py:313-316,386-388 under model.train()) in its
 * epilogue: part [tiles][2][Co] = per column tile (256 flattened (b, s) columns) the sum and the sum of squares of every
 * output channel - plain stores; eat_bn_finalize_partials(part, outer = tiles, C = Co, inner = 1) finishes.  The separate
 * statistics pass over z disappears.  wmode: 0 = eat_pw_prepack, 1 / 2 = eat_pw_prepack_bf16 (plain / split);
 * per_sample != 0: wp holds one pack per sample (eat_dyn_pw_pack / eat_dyn_pw_pack_bf16) and tiles lie inside samples;
 * tf_a / tf_b / tf_act and in_scale as eat_pw_conv_tf_fwd (NULL: none); zero_bias: Co zeros.
 * tiles = eat_pw_conv_stat_tiles(B, S, per_sample).  Returns 1 without launching where the epilogue does not exist
 * (S % 4 != 0): run eat_pw_conv_fwd + eat_bn_stats instead. */
int eat_pw_conv_stat_tiles(int B, int S, int per_sample);
int eat_pw_conv_stats_fwd(const float* x, const void* wp, int wmode, int per_sample, const float* tf_a, const float* tf_b,
                          int tf_act, const float* in_scale, const float* zero_bias, float* y, float* part, int B, int Ci,
                          int Co, int S, eat_stream_t stream);

/* Weight gradient of that conv: dW = sum dz (act_in(tf_a x + tf_b) * x_scale)^T with the transform on load. */
int eat_pw_conv_wgrad_tf(const float* dz, const float* x, const float* tf_a, const float* tf_b, int tf_act,
                         const float* x_scale, float* dW, float* ws, int n_slots, int B, int Co, int Ci, int S,
                         int exact_fp32, eat_stream_t stream);

/* ================= DyMN dynamic blocks (models/dymn/dy_block.py) ================================ */

/* ContextGen's two average pools (dy_block.py:236-237): x (B,C,F,T) -> seq (B, F+T, C), position-
 * major: rows 0..F-1 = mean over T, rows F..F+T-1 = mean over F. */
int eat_ctx_pool(const float* x, float* seq, int B, int C, int F, int T, eat_stream_t stream);

/* Kernel aggregation (dy_block.py:111-117): out[b,n] = gscale[n/group] * sum_k att[b,k]*bank[k,n];
 * bank (K,N) is DynamicConv.weight viewed as (K, Cout*Cin/g*k*k); gscale (N/group) or NULL folds the
 * eval-mode BatchNorm scale per output channel (group = Cin/g*k*k). */
int eat_dyn_aggregate(const float* bank, const float* att, const float* gscale, float* out, int B,
                      int K, int N, int group, eat_stream_t stream);

/* DynamicConv.forward, models/dymn/dy_block.py:111-119 (attention-weighted sum of the K kernels):
 * the same aggregation for a 1x1 DynamicConv written straight into MFMA A-fragment order, one
 * packed matrix of (Ci/4)*ceil(Co/16)*64 floats per sample (input of eat_pw_conv_dyn_fwd). */
int eat_dyn_pw_pack(const float* bank, const float* att, const float* row_scale, float* wp, int B,
                    int K, int Co, int Ci, eat_stream_t stream);

/* The same aggregation (models/dymn/dy_block.py:111-119) as split bf16 hi / lo MFMA fragments - eat_pw_prepack_bf16's
 * layout with split = 1, one pack of ceil(Ci/32)*ceil(Co/16)*2*512 bf16 per sample (input of eat_pw_conv_dyn_bf16_fwd). */
int eat_dyn_pw_pack_bf16(const float* bank, const float* att, void* wp, int B, int K, int Co, int Ci,
                         eat_stream_t stream);

/* Both packs from the bank of the TRANSPOSED matrices (bank_t (K, Ci*Co): W_k^T row-major = the parameter of the conv whose
 * data gradient is being formed, models/dymn/dy_block.py:111-127 backward): sum_k att[b,k] W_k (Co x Ci) without a
 * transposed copy of the bank.  Co % 4 == 0. */
int eat_dyn_pw_pack_t(const float* bank_t, const float* att, const float* row_scale, float* wp, int B, int K, int Co,
                      int Ci, eat_stream_t stream);
int eat_dyn_pw_pack_bf16_t(const float* bank_t, const float* att, void* wp, int B, int K, int Co, int Ci,
                           eat_stream_t stream);

/* 1x1 conv with per-sample weights (dy_block.py:120-127 for kernel_size 1): like eat_pw_conv_fwd
 * but sample b multiplies by wp_b[b]. */
int eat_pw_conv_dyn_fwd(const float* x, const float* wp_b, const float* bias, const float* res,
                        float* y, int B, int Ci, int Co, int S, int act, eat_stream_t stream);

/* ... and on split bf16 operands (bf16x3, fp32-class accuracy at a multiple of the fp32 MFMA rate; the grouped
 * F.conv2d of models/dymn/dy_block.py:120-127 for kernel_size 1): wp_b from eat_dyn_pw_pack_bf16.  Ci % 4 == 0, S % 4 == 0. */
int eat_pw_conv_dyn_bf16_fwd(const float* x, const void* wp_b, const float* bias, const float* res, float* y,
                             int B, int Ci, int Co, int S, int act, eat_stream_t stream);

/* Depthwise conv with per-(b,c) taps w_bc (B,C,k*k) + bias (C), followed in the epilogue by
 * DyReLU-B max(a1 v + b1, a2 v + b2) with coef (B,C,4) = (a1,a2,b1,b2) (dy_block.py:172-188) and
 * coordinate attention * sigmoid(gate_f[b,fo,c]) * sigmoid(gate_t[b,to,c]) (dy_block.py:195-201;
 * gates position-major (B,Fo,C) / (B,To,C), pre-sigmoid). */
int eat_dw_conv_dyn_fwd(const float* x, const float* w_bc, const float* bias, const float* coef,
                        const float* gate_f, const float* gate_t, float* y, int B, int C, int F, int T,
                        int Fo, int To, int k, int stride, eat_stream_t stream);
/* The same for the block's ablations (dy_block.py:353-356, DY_Block(no_dyrelu=..., no_ca=...)): `act` is the plain
 * activation that replaces DyReLU-B (dy_block.py:353: applied to the BN output), coef == NULL skips DyReLU-B, gate_f ==
 * gate_t == NULL skips the coordinate attention. */
int eat_dw_conv_dyn_act_fwd(const float* x, const float* w_bc, const float* bias, int act, const float* coef,
                            const float* gate_f, const float* gate_t, float* y, int B, int C, int F, int T,
                            int Fo, int To, int k, int stride, eat_stream_t stream);

/* ---- DyMN training step: backward of the dynamic pieces -------------------- */

/* autograd of the context pooling, models/dymn/dy_block.py:236-237 (adaptive_avg_pool2d over T and over F):
 * dx (B,C,F,T) = broadcast of dseq (B,F+T,C): dseq[b,f,c]/T + dseq[b,F+t,c]/F, plus add (same shape
 * as dx) or NULL. */
int eat_ctx_pool_bwd(const float* dseq, const float* add, float* dx, int B, int C, int F, int T,
                     eat_stream_t stream);

/* DyReLUB.forward (models/dymn/dy_block.py:172-188) + CoordAtt.forward (:195-201) and their autograd.
 * Stand-alone DyReLU-B + CoordAtt (train mode, after the BatchNorm statistics are known):
 * out = max(a1 v + b1, a2 v + b2) * sigmoid(gate_f) * sigmoid(gate_t), v = a_c z + b_c (a,b NULL: v=z);
 * the backward returns dv (same shape as z), dcoef (B,C,4), and the PRE-sigmoid gate gradients. */
int eat_dyrelu_ca_fwd(const float* z, const float* a, const float* b, const float* coef,
                      const float* gate_f, const float* gate_t, float* out, int B, int C, int Fo, int To,
                      eat_stream_t stream);
int eat_dyrelu_ca_bwd(const float* dout, const float* z, const float* a, const float* b, const float* coef,
                      const float* gate_f, const float* gate_t, float* dv, float* dcoef, float* dgate_f,
                      float* dgate_t, int B, int C, int Fo, int To, eat_stream_t stream);

/* Round 4: channel-major context generator of the training step (models/dymn/dy_block.py:235-254).
 * eat_ctx_pool_cm / _bwd: the two average pools (:236-237) with the sequence laid out seq (C, B, F+T) - the input of a
 * 1x1 conv over ONE sample of B*(F+T) positions, so that joint_conv / joint_norm / conv_f / conv_t (:238-252) run on the conv
 * and BatchNorm kernels of the feature maps.  eat_ctx_split / _bwd: g (H, B, F+T) after joint_norm + Hardswish ->
 * h_cf (H, B, Fo), h_ct (H, B, To) (the AvgPool(3, stride, pad 1) of :227-229 for stride 2) and h_c (B, H) = mean over L
 * (:244); the backward takes dh_c = NULL as zero. */
int eat_ctx_pool_cm(const float* x, float* seq, int B, int C, int F, int T, eat_stream_t stream);
int eat_ctx_pool_cm_bwd(const float* dseq, const float* add, float* dx, int B, int C, int F, int T, eat_stream_t stream);
int eat_ctx_split(const float* g, float* hcf, float* hct, float* hc, int H, int B, int F, int T, int stride,
                  eat_stream_t stream);
int eat_ctx_split_bwd(const float* dhcf, const float* dhct, const float* dhc, float* dg, int H, int B, int F, int T,
                      int stride, eat_stream_t stream);
/* Round 4 forms of DyReLU-B + CoordAtt (models/dymn/dy_block.py:172-188, :195-201; order inside DY_Block.forward :399-403)
 * for the training step: one wave per plane, To <= 512, gates channel-major and PRE-sigmoid, gate_f (C, B, Fo) / gate_t
 * (C, B, To) - the outputs of conv_f / conv_t on the channel-major context.  The backward returns dv, dcoef, the
 * pre-sigmoid gate gradients in the layouts of the gates and, when bnpart (B, C, 2) != NULL, the per-plane sums
 * (sum dv, sum dv * z) of the BatchNorm backward of depth_norm (:345-348). */
int eat_dyrelu_ca_fwd2(const float* z, const float* a, const float* b, const float* coef, const float* gate_f,
                       const float* gate_t, float* out, int B, int C, int Fo, int To, eat_stream_t stream);
int eat_dyrelu_ca_bwd2(const float* dout, const float* z, const float* a, const float* b, const float* coef,
                       const float* gate_f, const float* gate_t, float* dv, float* dcoef, float* dgate_f, float* dgate_t,
                       float* bnpart, int B, int C, int Fo, int To, eat_stream_t stream);
/* Channel sums of a train-mode BatchNorm backward (nn.BatchNorm2d autograd, models/dymn/dy_block.py:313-316,345-348) from
 * per-plane partial sums: sums[c] = sum p0, sums[C+c] = invstd[c] (sum p1 - mean[c] sum p0) in fp64 (the layout
 * eat_bn_act_bwd_apply / eat_dw_conv_dyn_bwd_bn_g read), dbeta = sums[0..C), dgamma = sums[C..2C) in fp32.  Element
 * (b, c, i) of a partial array lies stride_e * ((b*C + c)*inner + i) floats after its base pointer. */
int eat_bn_bwd_combine_partials(const float* p0, const float* p1, int stride_e, int B, int C, int inner, const float* mean,
                                const float* invstd, double* sums, float* dgamma, float* dbeta, eat_stream_t stream);

/* DyMN dynamic 1x1 conv without per-sample weights (models/dymn/dy_block.py:103-131, DynamicConv.forward with a 1x1
 * kernel): z_b = (sum_k att[b,k] W_k) x_b evaluated as ONE GEMM over the K-concatenated banks [W_0|...|W_{nbank-1}]
 * (wp = eat_pw_prepack_bf16 of the Co x (nbank*Ci) matrix, split form) with the attention as a per-(sample, k) scale of
 * the input (att_scale (B, nbank*Ci): att[b,k] repeated Ci times).  bf16x3 arithmetic (fp32-class).  Ci % 32 == 0,
 * S % 4 == 0.  The data gradient is the same call with the transposed banks. */
int eat_pw_conv_kcat_fwd(const float* x, const void* wp, const float* bias, const float* att_scale, const float* res,
                         float* y, int B, int Ci, int nbank, int Co, int S, int act, eat_stream_t stream);

/* autograd of the kernel aggregation of DynamicConv.forward (models/dymn/dy_block.py:103-127):
 * From the per-sample weight gradients G (B,N): dbank (K,N) = att^T G, datt (B,K) += G bank^T
 * (datt zeroed by the caller). */
int eat_dyn_bank_grad(const float* G, const float* att, const float* bank, float* dbank, float* datt,
                      int B, int K, int N, eat_stream_t stream);

/* Pointwise tails of the Linear layers that read the pooled context h_c of a DY_Block, one launch each way (round 6; the
 * reference evaluates them as separate modules: `F.softmax(self.residuals(g) / self.temperature)` per DynamicConv,
 * models/dymn/dy_block.py:106-109, and DyReLU-B's `2 * sigmoid(coef_net(g)) - 1`, `* lambdas + init_v`, :176-181):
 *   y (B, n_att*K + 4*cexp) - the row-concatenated Linear outputs (attention logits of the n_att <= 3 dynamic convs, then the
 *   4*cexp DyReLU coefficient pre-activations);  inv_t0..2 = 1 / temperature of each DynamicConv;
 *   att (n_att, B, K) = softmax over K;  sg (B, 4*cexp) = sigmoid (kept for the backward);
 *   coef (B, cexp, 4) = (2 sg - 1) * lambdas[m] + init_v[m].
 * eat_dyn_heads_bwd: dy (B, n_att*K + 4*cexp) from datt (n_att, B, K) and dcoef (B, cexp, 4). */
int eat_dyn_heads_fwd(const float* y, int B, int n_att, int K, int cexp, float inv_t0, float inv_t1, float inv_t2,
                      const float* lambdas, const float* init_v, float* att, float* sg, float* coef, eat_stream_t stream);
int eat_dyn_heads_bwd(const float* datt, const float* dcoef, const float* att, const float* sg, const float* lambdas, int B,
                      int n_att, int K, int cexp, float inv_t0, float inv_t1, float inv_t2, float* dy, eat_stream_t stream);

/* autograd of the grouped F.conv2d of DynamicConv.forward (models/dymn/dy_block.py:120-127):
 * Per-sample / per-plane variants of the conv gradients used by the dynamic convs: dW_b (B,Co,Ci)
 * (zeroed), dw_bc (B,C,k*k) (zeroed), and the depthwise data gradient with per-plane taps. */
int eat_pw_conv_dyn_wgrad(const float* dz, const float* x, float* dW_b, int B, int Co, int Ci, int S,
                          eat_stream_t stream);
/* host helper for the call above (models/dymn/dy_block.py:120-127 backward): 1 where it ADDS into dW_b (zero-fill it),
 * 0 where every element of dW_b is stored */
int eat_pw_dyn_wgrad_accumulates(int Co, int Ci, int S);
int eat_dw_conv_dyn_wgrad(const float* dz, const float* x, float* dw_bc, int B, int C, int F, int T,
                          int Fo, int To, int k, int stride, eat_stream_t stream);
int eat_dw_conv_dyn_dgrad(const float* dz, const float* w_bc, const float* res, float* dx, int B, int C,
                          int F, int T, int Fo, int To, int k, int stride, eat_stream_t stream);

/* Train-mode dynamic depthwise conv (models/dymn/dy_block.py:103-131 with groups = channels, under model.train()):
 * eat_dw_conv_fwd_stats with per-(b,c) taps w_bc (B, C, k*k) - the expand BatchNorm + activation (:313-318) evaluated on
 * load, the partial sums of depth_norm's batch statistics (:345) in the epilogue. */
int eat_dw_conv_dyn_fwd_stats(const float* x, const float* in_a, const float* in_b, int in_act, const float* w_bc, float* y,
                              float* part, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo, int To, int k,
                              int stride, eat_stream_t stream);
/* ... and its merged backward (autograd of :345-348 and :320-343 in one pass): eat_dw_conv_bwd_bn_g with per-plane taps.
 * dy = gradient w.r.t. the BatchNorm output of this conv, z its pre-BN output, sums from eat_bn_bwd_combine_partials;
 * g = dgrad * act_in'(in_a x + in_b) [+ res], dw_bc (B, C, k*k) zero-filled by the caller, gpart / gzpart: per-tile sums
 * of g and g * x (before res is added), inner_cap >= eat_dw_bwd_partials_inner(...) slots per plane. */
int eat_dw_conv_dyn_bwd_bn_g(const float* dy, const float* z, const float* bn_a, const float* bn_b, const float* bn_mean,
                             const float* bn_invstd, const double* sums, int bn_act, int frozen, const float* x,
                             const float* in_a, const float* in_b, int in_act, const float* w_bc, const float* res, float* g,
                             float* dw_bc, float* gpart, float* gzpart, int inner_cap, int* h_inner, int B, int C, int F,
                             int T, int Fo, int To, int k, int stride, eat_stream_t stream);

/* ---- fused expand 1x1 + depthwise k x k (eval): models/mn/block_types.py:138-162 (+ :72-73) ------
 * y (B,Cexp,Fo,To) = act(dw_k,s( act(W_e x + bias_e) ) + bias_d) without materialising the expanded
 * tensor: wp_e = eat_pw_prepack of the BN-folded expand weights, w_d (Cexp,k*k) BN-folded taps;
 * pool (B,Cexp) or NULL accumulates plane sums (zeroed by the caller).  Register-resident kernel (csrc/irb.hip) with
 * instantiations for the early-block geometries (ReLU; C_in 16 with 3x3 / stride 2, C_in 24 with 3x3 / stride 1 and
 * 5x5 / stride 2; Cexp % 8 == 0): ask eat_block_fused_supported, anything else returns EAT_EINVAL (use eat_pw_conv_fwd +
 * eat_dw_conv_fwd). */
int eat_fused_expand_dw_fwd(const float* x, const float* wp_e, const float* bias_e, const float* w_d,
                            const float* bias_d, float* y, float* pool, int B, int Cin, int Cexp, int F,
                            int T, int Fo, int To, int k, int stride, int act, eat_stream_t stream);

/* ---- training: batch-norm + activation of the expand conv fused into the depthwise conv -----------------
 * The depthwise conv of an inverted-residual block consumes act(BN(z_e)) (models/mn/block_types.py:138-162).
 * In the training step z_e and the per-channel affine (a, b) of its batch-norm are known, so the activated
 * tensor need not exist: the conv evaluates act_in(in_a[c] * x + in_b[c]) on load (zero padding applies to the
 * transformed map), and so does the weight gradient, which needs the same tensor as its x operand.
 *   eat_dw_conv_fwd_tf:   y (B,C,Fo,To) = dwconv_k,s(act_in(in_a x + in_b)) + bias, no output activation;
 *   eat_dw_conv_wgrad_tf: dw (C,k*k) ZEROED by the caller += sum dz * act_in(in_a x + in_b) (shifted). */
int eat_dw_conv_fwd_tf(const float* x, const float* in_a, const float* in_b, int in_act, const float* w,
                       const float* bias, float* y, int B, int C, int F, int T, int Fo, int To, int k,
                       int stride, eat_stream_t stream);
int eat_dw_conv_wgrad_tf(const float* dz, const float* x, const float* in_a, const float* in_b, int in_act,
                         float* dw, int B, int C, int F, int T, int Fo, int To, int k, int stride,
                         eat_stream_t stream);

/* ---- network front (eval): stem conv + first block in one kernel --------------------------------
 * models/mn/model.py:124-133 (3x3/s2 conv + BN + Hardswish on the (B,1,F,T) log-mel) followed by the first
 * inverted-residual block (no expand, no SE: depthwise 3x3/s1 + BN + act, project 1x1 + BN, + residual;
 * block_types.py:150-181).  w_s (C,9) / w_d (C,9) BN-folded taps, wp_p = eat_pw_prepack of the BN-folded
 * (C,C) project weights.  y (B,C,Fo,To), Fo = (F-1)/2+1, To = (T-1)/2+1.  C must be 16. */
int eat_front_fwd(const float* x, const float* w_s, const float* bias_s, const float* w_d,
                  const float* bias_d, const float* wp_p, const float* bias_p, float* y, int B, int C, int F,
                  int T, int Fo, int To, int act, eat_stream_t stream);

/* ---- whole inverted-residual block without SE (eval): models/mn/block_types.py:138-181 -----------
 * y (B,Cout,Fo,To) = W_p . act(dw_k,s( act(W_e x + bias_e) ) + bias_d) + bias_p [+ res]; neither the
 * expanded tensor nor the depthwise output touches HBM.  wp_e / wp_p = eat_pw_prepack of the BN-folded
 * expand (Cexp,Cin) / project (Cout,Cexp) weights, w_d (Cexp,k*k) BN-folded taps.  res (B,Cout,Fo,To)
 * or NULL is the residual input (stride 1, Cin == Cout).  Instantiated for mn10's blocks 2 and 3 (16 -> 64 -> 24 with
 * 3x3 / stride 2, 24 -> 72 -> 24 with 3x3 / stride 1, ReLU); eat_block_fused_supported answers for a geometry, anything
 * else returns EAT_EINVAL (use the separate kernels). */
int eat_mbconv_fwd(const float* x, const float* wp_e, const float* bias_e, const float* w_d,
                   const float* bias_d, const float* wp_p, const float* bias_p, const float* res, float* y,
                   int B, int Cin, int Cexp, int Cout, int F, int T, int Fo, int To, int k, int stride,
                   int act, eat_stream_t stream);

/* Host query (no launch): 1 when eat_mbconv_fwd (proj = 1) / eat_fused_expand_dw_fwd (proj = 0) has an instantiation for
 * an InvertedResidual block (models/mn/block_types.py:138-181) of this geometry on an (F, T) input plane, else 0. */
int eat_block_fused_supported(int Cin, int Cexp, int Cout, int F, int T, int k, int stride, int act, int proj);

/* Expand 1x1 conv + BN + act -> depthwise 3x3 (stride 1) conv + BN + act [+ SE squeeze sums] in one kernel for small
 * planes (models/mn/block_types.py:138-162, :72-73): the expanded tensor stays in LDS (csrc/expand_dw.hip).
 * x (B,Ci,F,T) -> y (B,Ce,F,T); wp_e = eat_pw_prepack_bf16(split = 1) of the BN-folded expand weights (bf16x3
 * arithmetic), bias_e (Ce); w_d (Ce,9), bias_d (Ce) BN-folded depthwise taps; pool (B,Ce) accumulates plane sums of y,
 * or NULL.  Supported: k = 3, stride = 1, T <= 64, F <= 8, F*T % 4 == 0, F*T <= 512, Ci % 4 == 0, Ci <= 128; anything
 * else returns EAT_EINVAL (use eat_pw_conv_bf16_fwd + eat_dw_conv_fwd). */
int eat_expand_dw_bf16_fwd(const float* x, const void* wp_e, const float* bias_e, const float* w_d,
                           const float* bias_d, float* y, float* pool, int B, int Ci, int Ce, int F, int T,
                           int k, int stride, int act, eat_stream_t stream);

/* ---- 1x1 conv on the bf16 matrix cores (fp32 activations in memory, fp32 accumulation) -----------
 * split != 0: "bf16x3" - x and w are split into bf16 hi + lo parts and y = w_hi x_hi + w_hi x_lo +
 *             w_lo x_hi (~2^-16 relative error per product) at 3/16 of the fp32-MFMA time;
 * split == 0: plain bf16 operands (BASELINE config 3, bf16 compute).
 * wp from eat_pw_prepack_bf16: ceil(Ci/32)*ceil(Co/16)*(split?2:1)*512 bf16 values.  Other arguments
 * as eat_pw_conv_fwd. */
int eat_pw_prepack_bf16(const float* w, const float* row_scale, void* wp, int Co, int Ci, int split,
                        eat_stream_t stream);
/* bf16 pack of the transpose of a stored (Ci, Co) matrix, see eat_pw_prepack_t. */
int eat_pw_prepack_bf16_t(const float* w_t, const float* row_scale, void* wp, int Co, int Ci, int split,
                          eat_stream_t stream);
int eat_pw_conv_bf16_fwd(const float* x, const void* wp, const float* bias, const float* in_scale,
                         const float* res, float* y, float* pool, int B, int Ci, int Co, int S, int act,
                         int split, eat_stream_t stream);
/* Data-movement variant behind eat_pw_conv_bf16_fwd (same arithmetic; results agree to fp32 round-off of the bias add):
 * bit 0 = expand-shaped layers (C_out >= 2 C_in, C_in <= 128, no SE scale / residual / pool) on the x-resident kernel,
 * bit 1 = the project-shaped layers on which it measured faster on the K-streaming kernel, bit 2 = every remaining layer
 * on the K-streaming kernel, bit 3 = the K-concat launches of eat_pw_conv_kcat_fwd on it (csrc/conv_pw_stream.hip);
 * 0 = every layer on the LDS-staged kernel.  mode >= 0 sets it process-wide (not per stream; meant for A/B runs and
 * tests), mode < 0 only queries.  Returns the mode in effect BEFORE the call.  Default: environment variable
 * EAT_PW_STREAM, else 2. */
int eat_pw_stream_mode(int mode);

/* ================= training-loop glue (ex_audioset.py:142-194) ========================= */

/* Column sums of a row-major (R, C) matrix: out[c] = sum_r m[r, c] (out is overwritten).  Replaces the `dy.sum(0)` bias
 * gradients autograd computes for the context-path Linear / 1x1-conv layers of DyMN (models/dymn/dy_block.py:235-254,
 * `conv_f`, `conv_t`, `joint_conv` biases; ex_audioset.py:150-153 `loss.backward()`). */
int eat_col_sum(const float* m, float* out, int R, int C, eat_stream_t stream);

/* Measurement support (bench.py --calibrate-traffic; not on the hot path - the reference has no counterpart): copy n
 * floats (n % 4 == 0) with a chosen access width, mode 0 = 16 B / lane global loads, 1 = 16 B / lane LDS-DMA loads,
 * 2 = 4 B / lane, 3 = 8 B / lane.  4 n bytes are read and 4 n written: the known byte count against which the
 * rocprofv3 FETCH_SIZE / WRITE_SIZE readings of the same pass are calibrated (gfx950 reports wide reads at half size). */
int eat_calib_copy(const float* src, float* dst, long long n, int mode, eat_stream_t stream);

/* Mix-up of a batch with itself (ex_audioset.py:142-148, helpers/utils.py:90-95): out[b] = x[b]*lam[b] + x[perm[b]]*(1-lam[b]);
 * x, out (B, n) fp32 (n = the flattened per-sample size), perm (B) int32, lam (B) fp32.  out must not alias x. */
int eat_mixup_fwd(const float* x, const int* perm, const float* lam, float* out, int B, int n, eat_stream_t stream);

/* 16-bit PCM transport of the waveforms (the reference moves fp32 clips with a blocking x.to(device),
 * ex_audioset.py:140-141, 303-304 - 1.28 MB per clip, more than PCIe carries at this path's rates): dst[i] = src[i] * scale
 * for n samples (scale = 1 / 32768 for full-scale PCM).  src int16, dst fp32, both 16-byte aligned. */
int eat_wave_i16_to_f32(const short* src, float* dst, long long n, float scale, eat_stream_t stream);

/* Loss of the KD training step and its gradient w.r.t. the logits in one pass (ex_audioset.py:149-189):
 *   label = mean_{b,c} BCEwithLogits(z, y*lam + y[perm]*(1-lam))
 *   kd    = mean_b known[b] * ( lam[b] * mean_c BCE(z, t[idx[b]]) + (1-lam[b]) * mean_c BCE(z, t[idx[perm[b]]]) )
 *   loss  = kd_lambda * label + (1 - kd_lambda) * kd              (kd_lambda in [0,1]; teacher == NULL: loss = label)
 * logits, y, dlogits (B, C); perm / lam (B) or both NULL (no mix-up); teacher (n_teacher, C) PROBABILITIES
 * (sigmoid(teacher_logits / temperature), as the reference stores them), teacher_idx (B) int64 with -1 for files that
 * have no teacher entry (their KD term is zeroed; like the reference, row n_teacher-1 stands in for the lookup).
 * sums (3) fp32 += (loss, kd_lambda*label, (1-kd_lambda)*kd): accumulate over steps on the device, read once per epoch
 * (the reference syncs three scalars to the host every step, ex_audioset.py:192-194).  dlogits = d loss / d logits. */
int eat_kd_loss_fwd_bwd(const float* logits, const float* y, const int* perm, const float* lam, const float* teacher,
                        const long long* teacher_idx, int n_teacher, float kd_lambda, int B, int C, float* sums,
                        float* dlogits, eat_stream_t stream);

/* The optimizer step of the training loop (ex_audioset.py:86-91 `torch.optim.Adam` / `AdamW` over all parameters, :197-199
 * `optimizer.step()`) as ONE launch over every parameter: table = n_chunks records of 40 bytes
 * { float* p; const float* g; float* m; float* v; int n; int pad; } - a chunk is <= 4096 consecutive elements of one parameter
 * with its gradient and the two moment buffers (fp32, updated in place).
 *   g' = g * grad_scale (+ weight_decay * p when decoupled == 0: Adam's L2 form);  decoupled != 0 (AdamW): p *= 1 - lr * weight_decay;
 *   m += (1 - beta1) (g' - m);  v = beta2 v + (1 - beta2) g'^2;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * with t = step + 1.  lr_ptr / step_ptr != NULL: the learning rate / step counter are read from the device (hipGraph replays,
 * schedulers that write a tensor) and *step_ptr is advanced by one after the update; NULL: the by-value arguments are used.
 * Hyper-parameters are doubles and the second-moment / update expressions are evaluated in fp64 where torch's fused kernel
 * evaluates them in fp64 (aten/src/ATen/native/cuda/fused_adam_utils.cuh: double hyper-parameters promote those lines). */
int eat_adam_multi(const void* table, int n_chunks, const float* lr_ptr, double lr, float* step_ptr, float step, double beta1,
                   double beta2, double eps, double weight_decay, int decoupled, double grad_scale, eat_stream_t stream);

/* Backward of the classifier head (models/mn/model.py:186-194: Linear(C -> H) -> Hardswish -> Dropout -> Linear(H -> N); the
 * reference leaves it to autograd) in two launches, no transposed copies, every output element from one block (no atomics):
 *   dW2 = dlogits^T h2, db2 = sum_b dlogits, du = (dlogits W2) * drop_mask * hardswish'(u), dW1 = du^T feat, db1 = sum_b du,
 *   dfeat = du W1.  dlogits (B, N); h2 (B, H) the second Linear's input (after dropout); u (B, H) the first Linear's output;
 *   drop_mask (B, H) keep / (1 - p) factors or NULL; feat (B, C); W1 (H, C); W2 (N, H); du (B, H) scratch; dfeat:
 *   eat_mlp_head_dfeat_floats(B, C, H) floats, the result (B, C) first (long contractions are split and summed in a fixed
 *   order). */
int eat_mlp_head_dfeat_floats(int B, int C, int H);
int eat_mlp_head_bwd(const float* dlogits, const float* h2, const float* u, const float* drop_mask, const float* feat,
                     const float* W1, const float* W2, float* dW1, float* db1, float* dW2, float* db2, float* du,
                     float* dfeat, int B, int C, int H, int N, eat_stream_t stream);

/* ==== bf16 ACTIVATION STORAGE (BASELINE configs[2]: "mn40_as train step bf16") ============================================
 * The reference trains in 16-bit mixed precision through PyTorch-Lightning (`precision=16`, ex_pl_audioset.py:287-293;
 * the same surface as torch.autocast around models/mn/model.py:212-231): conv outputs and the gradients flowing back
 * through them are stored in 16 bits, BatchNorm statistics, parameters, their gradients and the optimizer state in fp32.
 * Here that is the `_b16` family: the WIDE tensors of an inverted-residual block (models/mn/block_types.py:138-181) - the
 * expand conv's output z_e, the depthwise output z_d, its activated form y_d, and the gradients dxs / g arriving at them -
 * are bf16 in HBM (`const void*` = bf16 where the matching `*_b16` flag says so); the narrow block inputs / outputs and
 * every per-channel quantity stay fp32; arithmetic is fp32 in registers, GEMM operands plain bf16 with fp32 accumulation.
 * Stores round to nearest even; statistics are taken of the values AS STORED.  Planes must hold an even number of elements
 * (bf16 planes then start on 4-byte boundaries).  Each entry point below is the bf16-storage twin of the fp32 one named in
 * its comment and replaces the same reference call site. */

/* Twin of eat_pw_conv_fwd / _tf_fwd / _stats_fwd / _cat_fwd (models/mn/block_types.py:138-147,167-181): exactly one of x / y is
 * the wide bf16 tensor.
 *   x_b16 = 0, y_b16 = 1: z_e = W x (expand conv) or dxs = Wp^T dz_p (project data gradient): plain conv, everything optional NULL;
 *   x_b16 = 1, y_b16 = 0: z_p = Wp (act(tf_a x + tf_b) * in_scale) with the statistics epilogue (stats_part as
 *          eat_pw_conv_stats_fwd, tiles = eat_pw_conv_stat_tiles(B, S, 0)), or the two-source data-gradient GEMM
 *          dx = [WaT | M] [g ; x2] + bias + res with x2 (B, Ci - c1, S) fp32, c1 % 32 == 0 (as eat_pw_conv_cat_fwd);
 *   x_b16 = 1, y_b16 = 1: the project conv with its output z_p stored in bf16 too (statistics of the stored values); the
 *          expand conv / project data gradient from a bf16 copy of the narrow operand (eat_cast_b16); with gz != NULL
 *          (bf16 z_d, B x Co x S) the project data gradient with the depthwise BatchNorm's backward sums in its epilogue:
 *          stats_part then holds the partials of eat_pw_conv_gstats_fwd (g formed from y AS STORED).
 * wp = eat_pw_prepack_bf16(split = 0) of the (Co, Ci) matrix, Ci = all reduction channels; S % 8 == 0, Ci % 4 == 0
 * (% 8 with a transform). */
int eat_pw_conv_b16_fwd(const void* x, int x_b16, const float* x2, int c1, const void* wp, const float* bias,
                        const float* tf_a, const float* tf_b, int tf_act, const float* in_scale, const float* res, void* y,
                        int y_b16, float* stats_part, const void* gz, const float* g_a, const float* g_b, int g_act, int B,
                        int Ci, int Co, int S, int act, eat_stream_t stream);

/* Twin of eat_dw_conv_fwd_stats (models/mn/block_types.py:150-162 under model.train()): y bf16, x bf16 (x_b16 != 0) or fp32
 * (the first block's depthwise conv reads the stem output); partial sums of the rounded outputs.  eat_dw_conv_b16_ok(...) != 0 where this and eat_dw_conv_bwd_bn_g_b16 cover the geometry (a plan keeps fp32
 * storage for the other blocks). */
int eat_dw_conv_b16_ok(int B, int C, int F, int T, int Fo, int To, int k, int stride);
int eat_dw_conv_fwd_stats_b16(const void* x, int x_b16, const float* in_a, const float* in_b, int in_act, const float* w, void* y,
                              float* part, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo, int To, int k,
                              int stride, eat_stream_t stream);

/* bf16 copy (round to nearest even) of n fp32 values, n % 8 == 0: the narrow operand of the widest blocks' expand /
 * data-gradient 1x1 convs (models/mn/block_types.py:127-133), which eat_pw_conv_b16_fwd then reads with x_b16 = y_b16 = 1 -
 * bit-identical to handing it the fp32 tensor (the kernel rounds the same way), at half the operand traffic. */
int eat_cast_b16(const float* x, void* y, long long n, eat_stream_t stream);

/* Twin of eat_bn_act_fwd (block_types.py:150-162, 72-73, 167-181): z bf16 -> y bf16 (y_b16 != 0; or NULL: squeeze sums only)
 * or -> y fp32 with the optional fp32 residual `res` (the project conv's BatchNorm: z_p stored in bf16, the block output in
 * fp32) and, optionally, y_copy16 = the bf16 rounding of that fp32 y written by the same pass (what the next block's expand
 * conv reads: see eat_cast_b16); pool (B, C) plain stores of the sums of y as stored. */
int eat_bn_act_fwd_b16(const void* z, const float* a, const float* b, const float* res, void* y, int y_b16, void* y_copy16,
                       float* pool, int B, int C, int S, int act, eat_stream_t stream);

/* Twins of eat_bn_act_bwd_reduce / _apply and eat_se_bn_bwd_partials (backward of block_types.py:150-181, 72-83): z bf16; dy
 * bf16 (dy_b16 != 0: the depthwise BatchNorm, gradient dxs) or fp32 (the project BatchNorm); the apply pass (project
 * BatchNorm only) reads fp32 dy and writes fp32 dz and, optionally, dz_copy16 = its bf16 rounding (what the data-gradient
 * 1x1 conv reads). */
int eat_bn_act_bwd_reduce_b16(const void* dy, int dy_b16, const void* z, const float* a, const float* b, const float* mean,
                              const float* invstd, const float* gscale, const float* gadd, int B, int C, int S, int act,
                              double* sums, eat_stream_t stream);
int eat_bn_act_bwd_apply_b16(const float* dy, const void* z, const float* a, const float* b, const float* mean,
                             const float* invstd, const float* gscale, const float* gadd, const double* sums, float* dz,
                             void* dz_copy16, int B, int C, int S, int act, eat_stream_t stream);
int eat_se_bn_bwd_partials_b16(const void* d, const void* z, const float* a, const float* b, const float* mean, float* P,
                               int B, int C, int S, int act, eat_stream_t stream);

/* Twin of eat_dw_conv_bwd_bn_g (backward of block_types.py:138-162): dy and z are bf16; x and the output g are bf16
 * (x_b16 != 0) or both fp32 (the first block). */
int eat_dw_conv_bwd_bn_g_b16(const void* dy, const void* z, const float* bn_a, const float* bn_b, const float* bn_mean,
                             const float* bn_invstd, const float* gscale, const float* gadd, const double* sums, int bn_act,
                             int frozen, const void* x, int x_b16, const float* in_a, const float* in_b, int in_act, const float* w,
                             void* g, float* dw, float* gpart, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo,
                             int To, int k, int stride, eat_stream_t stream);

/* Twin of eat_pw_conv_wgrad_ws / _tf (backward of block_types.py:138-147,167-181): dW (Co, Ci) += sum_b dz[b] x'[b]^T with
 * exactly one bf16 operand - x (then x' = act(tf_a x + tf_b) * x_scale, each optional) or dz.  ws: workspace of
 * n_slots >= eat_pw_wgrad_b16_slots(B, Co, Ci, S, x_b16) copies of dW (no zero fill needed); dW is added to.
 * S % 4 == 0, Ci % 4 == 0. */
int eat_pw_wgrad_b16_slots(int B, int Co, int Ci, int S, int x_b16);
int eat_pw_conv_wgrad_b16(const void* dz, int dz_b16, const void* x, int x_b16, const float* tf_a, const float* tf_b,
                          int tf_act, const float* x_scale, float* dW, float* ws, int n_slots, int B, int Co, int Ci, int S,
                          eat_stream_t stream);

/* ---- bf16 activation storage for the DyMN blocks (BASELINE configs[3] on the SURVEY 8(d) byte contract, which is quoted for
 * bf16 activations; the reference's 16-bit surface is the same Lightning `precision=16`, ex_pl_audioset.py:287-293, over
 * models/dymn/dy_block.py:390-409).  The wide tensors of a DY_Block - the dynamic expand conv's output z_e, the dynamic
 * depthwise output z_d, the DyReLU-B * CoordAtt output, and the gradients arriving at them - are bf16 in HBM; block inputs /
 * outputs, the context path, coefficients, gates, per-sample weights' gradients and every statistic stay fp32.  GEMM operands
 * are plain bf16 with fp32 accumulation.  Each entry point is the bf16-storage twin of the fp32 one named in its comment. */

/* Twin of eat_dyn_pw_pack_bf16 / _t (models/dymn/dy_block.py:111-119): the aggregated per-sample weights as PLAIN bf16
 * fragments (eat_pw_prepack_bf16's layout with split = 0): ceil(Ci/32)*ceil(Co/16)*512 bf16 per sample.  trans != 0: `bank`
 * holds the transposed matrices (K, Ci*Co) - the data-gradient pack; then Co % 4 == 0.  Ci % 4 == 0. */
int eat_dyn_pw_pack_b16(const float* bank, const float* att, void* wp, int B, int K, int Co, int Ci, int trans,
                        eat_stream_t stream);

/* Twin of eat_pw_conv_dyn_bf16_fwd / eat_pw_conv_stats_fwd(per_sample = 1) (dy_block.py:120-127 for kernel_size 1, forward and
 * data gradient): exactly one of x / y is the wide bf16 tensor.
 *   x_b16 = 0, y_b16 = 1: z_e = W_b x (dynamic expand conv) or dx2 = W_b^T dz_p (project data gradient); no residual;
 *   x_b16 = 1, y_b16 = 0: z_p = W_b x2 (dynamic project conv) or dx = W_b^T dz_e + res (expand data gradient).
 * stats_part != NULL: the batch statistics of y AS STORED in the epilogue ([tiles][2][Co], tiles =
 * eat_pw_conv_stat_tiles(B, S, 1); finish with eat_bn_finalize_partials).  wp_b: eat_dyn_pw_pack_b16; bias (Co) - zeros for a
 * conv without bias.  S % 8 == 0, Ci % 4 == 0. */
int eat_pw_conv_dyn_b16_fwd(const void* x, int x_b16, const void* wp_b, const float* bias, const float* res, void* y,
                            int y_b16, float* stats_part, int B, int Ci, int Co, int S, int act, eat_stream_t stream);

/* Twin of eat_dw_conv_dyn_fwd_stats (dy_block.py:103-131 with groups = channels, under model.train()): y bf16; x bf16, or fp32
 * with x_b16 = 0 (the block without expand conv; tile geometries, T > 128, only).  Planes hold an even number of elements;
 * geometries: eat_dw_conv_b16_ok. */
int eat_dw_conv_dyn_fwd_stats_b16(const void* x, int x_b16, const float* in_a, const float* in_b, int in_act, const float* w_bc,
                                  void* y, float* part, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo, int To,
                                  int k, int stride, eat_stream_t stream);

/* Twins of eat_dyrelu_ca_fwd2 / _bwd2 (dy_block.py:172-188, 195-201; order :399-403): z, out, dout and dv are bf16; the outputs
 * are rounded on store and bnpart holds the sums of dv as stored. */
int eat_dyrelu_ca_fwd2_b16(const void* z, const float* a, const float* b, const float* coef, const float* gate_f,
                           const float* gate_t, void* out, int B, int C, int Fo, int To, eat_stream_t stream);
int eat_dyrelu_ca_bwd2_b16(const void* dout, const void* z, const float* a, const float* b, const float* coef,
                           const float* gate_f, const float* gate_t, void* dv, float* dcoef, float* dgate_f, float* dgate_t,
                           float* bnpart, int B, int C, int Fo, int To, eat_stream_t stream);

/* Twin of eat_dw_conv_dyn_bwd_bn_g (backward of dy_block.py:320-348): dy and z are bf16; x and g are bf16 (x_b16 != 0), or
 * both fp32 with the optional fp32 skip gradient res (x_b16 = 0: the block without expand conv, 3x3 / stride 1 on planes wider
 * than 128 columns only). */
int eat_dw_conv_dyn_bwd_bn_g_b16(const void* dy, const void* z, const float* bn_a, const float* bn_b, const float* bn_mean,
                                 const float* bn_invstd, const double* sums, int bn_act, int frozen, const void* x, int x_b16,
                                 const float* in_a, const float* in_b, int in_act, const float* w_bc, const float* res, void* g,
                                 float* dw_bc, float* gpart, float* gzpart, int inner_cap, int* h_inner, int B, int C, int F,
                                 int T, int Fo, int To, int k, int stride, eat_stream_t stream);

/* Twin of eat_bn_act_bwd_apply for the expand BatchNorm of a DY_Block (backward of dy_block.py:313-318): dy (= g_e, which
 * already carries the activation derivative: act = 0), z (= z_e) and dz (= dz_e; may alias dy) are all bf16. */
int eat_bn_bwd_apply_b16(const void* dy, const void* z, const float* a, const float* b, const float* mean, const float* invstd,
                         const double* sums, void* dz, int B, int C, int S, int act, eat_stream_t stream);

/* Twin of eat_pw_conv_dyn_wgrad (autograd of the grouped F.conv2d of dy_block.py:120-127): per-sample weight gradients
 * dW_b (B, Co, Ci) = dz[b] x[b]^T with exactly one bf16 operand; every element is stored (no zero fill).  dW_b holds
 * n_slices >= eat_pw_dyn_wgrad_b16_slices(...) copies of (B, Co, Ci) floats: the result is the FIRST copy, the others are
 * workspace (the reduction of a sample is cut into k-slices where B x tiles alone would not fill the chip; the slices are
 * added in a fixed order).  S % 4 == 0, Ci % 4 == 0.  dz_b16 = x_b16 = 0 (round 6): both operands fp32 - the per-sample
 * gradients of the fp32-storage plan on the same kernel with split-operand (bf16x3) products (eat_pw_conv_dyn_wgrad's
 * arithmetic; 1.3 - 2.6x faster than its per-(tile, sample) kernels).  eat_pw_dyn_wgrad_b16_slices: x_b16 = 1 / 0 as in the
 * call, 2 = both fp32; returns 0 where the wide-tile kernel does not take the shape (use eat_pw_conv_dyn_wgrad). */
int eat_pw_dyn_wgrad_b16_slices(int B, int Co, int Ci, int S, int x_b16);
int eat_pw_conv_dyn_wgrad_b16(const void* dz, int dz_b16, const void* x, int x_b16, float* dW_b, int n_slices, int B, int Co,
                              int Ci, int S, eat_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* EAT_HIP_H */
