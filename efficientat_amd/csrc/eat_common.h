// Shared helpers for the gfx950 kernels of libeat_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/eat_hip.h"

namespace eat {

// thread-local last-error text behind eat_last_error_string()
char* err_buf();
int fail(int code, const char* fmt, ...);

// hipGetLastError is sticky per thread: drop whatever an earlier, unrelated runtime call left behind
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EAT_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return EAT_OK;
}

template <int ACT>
__device__ __forceinline__ float activate(float v) {
  if constexpr (ACT == EAT_ACT_RELU) return fmaxf(v, 0.0f);
  if constexpr (ACT == EAT_ACT_HSWISH) return v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f);
  return v;
}

__device__ __forceinline__ float activate_rt(float v, int act) {
  if (act == EAT_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == EAT_ACT_HSWISH) return v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f);
  return v;
}

// 64-lane wavefront sum (all lanes receive the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int kWave = 64;

// A value the optimiser cannot see through (no instruction is emitted).  Used for 0 / 1 validity factors: LLVM rewrites
// `x * (cond ? 1 : 0)` into `cond ? x : 0` and then sinks the loads x depends on into an exec-masked block with its own
// s_waitcnt - one exposed memory latency per element instead of one per batch of loads.
// (not `volatile`: a volatile asm is a scheduling boundary and would itself keep loads from being batched)
__device__ __forceinline__ float opaque(float v) {
  asm("" : "+v"(v));
  return v;
}
__device__ __forceinline__ float opaque_uniform(float v) {       // wave-uniform value, stays in a scalar register
  asm("" : "+s"(v));
  return v;
}

// Training epilogues of the register-resident depthwise kernels (dw_plane.hip), all optional:
//   stats   forward: per-wave partial sums of the conv output, floats [b][2][C][inner] (sum, sum of squares) - the
//           BatchNorm batch statistics without a separate pass over the tensor (eat_bn_finalize_partials reduces them);
//   gz/ga/gb/gact/gpart   data gradient: the output dx is multiplied by act'(ga[c] * gz + gb[c]) (gz: the pre-BN tensor
//           of the layer below, same shape as dx) and summed per wave into gpart [b][C][inner];
//   inner   (host pointer) receives the number of partial slots per plane of the kernel that ran.
struct DwEpi {
  float* stats;
  const float* gz; const float* ga; const float* gb; int gact; float* gpart;
  int* inner;
};

// conv_spatial.hip: stride-1 depthwise data gradient on the forward sliding-window kernel
// (epi with gz: the training epilogue; returns 1 when no register-resident kernel covers the geometry)
int dw_conv_dgrad_s1(const float* dz, const float* w, const float* zero_bias, const float* res, float* dx, int B, int C,
                     int F, int T, int k, int per_plane_w, hipStream_t s, const DwEpi* epi = nullptr);

// train_fuse.hip: generic (any geometry) forms of the two training epilogues, one block per plane, inner = 1
int bn_stats_partial(const float* z, int B, int C, int S, float* part, hipStream_t s);
int act_grad_sum(const float* dy, const float* z, const float* a, const float* b, int act, float* g, float* gpart, int B,
                 int C, int S, hipStream_t s);

// conv_pw_generic.hip: 1x1 conv for plane sizes that are not a multiple of 4 (same packed weights / epilogue contract;
// wmode 0 = fp32 pack, 1 = bf16 pack, 2 = bf16 hi/lo pack; wp_bstride_bytes != 0 selects per-sample weights)
int pw_conv_generic(const float* x, const void* wp, const float* bias, const float* in_scale, const float* res, float* y,
                    float* pool, int B, int Ci, int Co, int S, int act, int wmode, long long wp_bstride_bytes,
                    hipStream_t s);

// conv_pw_bf16.hip: 1x1 conv on the bf16 packs whose input is act_in(tf_a[k] x + tf_b[k]) evaluated on load
int pw_conv_bf16_tf(const float* x, const float* tf_a, const float* tf_b, int tf_act, const void* wp, const float* bias,
                    const float* in_scale, const float* res, float* y, int B, int Ci, int Co, int S, int act, int split,
                    hipStream_t s);

// BatchNorm-backward statistics in a 1x1 conv's epilogue (pw_epilogue.h: pw_epilogue_gstats): z = the tensor whose
// BatchNorm + activation backward is being reduced (layout of the conv output), a / b = that BatchNorm's folded affine
struct PwGStat { const void* z; const float* a; const float* b; int act; };
// centring constant of the BatchNorm-backward partials of pw_epilogue_gstats (pw_epilogue.h) and of bn_bwd_sums_finish_kernel
// (train_fuse.hip), from the BatchNorm's (a, b): the zero of the pre-activation.  ONE expression, so that the epilogue subtracts
// and the finish adds back the same fp32 number.
__host__ __device__ __forceinline__ float gstat_center(float a, float b) { return a != 0.0f ? -b / a : 0.0f; }

int pw_conv_bf16_stats(const float* x, const void* wp, int split, int per_sample, const float* tf_a, const float* tf_b,
                       int tf_act, const float* in_scale, const float* zero_bias, float* y, float* part, int B, int Ci, int Co,
                       int S, hipStream_t s, PwGStat gs = PwGStat{nullptr, nullptr, nullptr, 0});

int pw_conv_bf16_cat(const float* x1, int c1, const float* x2, int c2, const void* wp, const float* bias, const float* res,
                     float* y, int B, int Co, int S, int act, int split, hipStream_t s);

// conv_pw_stream.hip: barrier-free bf16 1x1 kernels (x or the output tile resident in registers); returns 1 when the
// shape / the EAT_PW_STREAM switch leaves the layer to conv_pw_bf16.hip
int pw_stream_try(const float* x, const void* wp, const float* bias, const float* in_scale, const float* res, float* y,
                  float* pool, int B, int Ci, int Co, int S, int act, int split, int ci_x, hipStream_t s);

int dw_plane_try(const float* x, const float* w, const float* bias, const float* res, float* y, float* pool, int B,
                 int C, int F, int T, int Fo, int To, int k, int stride, int act, int flip, int per_plane_w, const float* in_a,
                 const float* in_b, int in_act, hipStream_t s, const DwEpi* epi = nullptr, int b16 = 0);
int dw_plane_wgrad_try(const float* dz, const float* x, float* dw, int B, int C, int F, int T, int Fo, int To, int k,
                       int stride, int per_plane, const float* in_a, const float* in_b, int in_act, hipStream_t s);
// merged depthwise backward (dw_plane.hip): weight gradient + data gradient + activation-derivative epilogue in one pass;
// returns 1 when switched off (EAT_DW_BWD_MERGED=0) or the geometry is out of range
// bn != NULL: dz is the gradient w.r.t. the activated BatchNorm output of this conv; the BatchNorm + activation backward is
// evaluated on load from (dz, bn->z) with the channel sums of the reduce pass (sums[0..C) = sum g, [C..2C) = sum g xhat)
struct DwBnBwd {
  const float* z; const float* a; const float* b; const float* mean; const float* invstd;
  const float* gscale; const float* gadd; const double* sums; int act; int frozen;
};
int dw_bwd_try(const float* dz, const float* x, const float* in_a, const float* in_b, int in_act, const float* w, float* g,
               float* dw, float* gpart, int* h_inner, int B, int C, int F, int T, int Fo, int To, int k, int stride,
               hipStream_t s, const DwBnBwd* bn = nullptr, int per_plane_w = 0, const float* res = nullptr,
               float* gzpart = nullptr, int b16 = 0);
// b16 (dw_plane_try, dw_bwd_try): the wide tensors (x and y; dz, bn->z, x and g) are bf16 in HBM (act_io.h) and the pointers
// are really bf16_t*: the statistics forward / the BatchNorm-on-load backward instances only, 1 = geometry not covered
int dw_tile_dgrad2_try(const float* dz, const float* w, const float* res, float* dx, int B, int C, int F, int T, int Fo,
                       int To, int k, int per_plane_w, hipStream_t s, const DwEpi* epi = nullptr);

}  // namespace eat

#define EAT_DISPATCH_ACT(act, ...)                                    \
  do {                                                                \
    if ((act) == EAT_ACT_NONE) { constexpr int ACT = EAT_ACT_NONE; __VA_ARGS__; }        \
    else if ((act) == EAT_ACT_RELU) { constexpr int ACT = EAT_ACT_RELU; __VA_ARGS__; }   \
    else { constexpr int ACT = EAT_ACT_HSWISH; __VA_ARGS__; }                            \
  } while (0)
