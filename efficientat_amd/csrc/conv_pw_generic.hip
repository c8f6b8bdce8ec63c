// 1x1 convolution for plane sizes the MFMA kernels cannot tile (S = F*T not a multiple of 4).
//   reference: the same call sites as conv_pw.hip (models/mn/block_types.py:138-147,167-171,83,177-181)
//
// conv_pw.hip / conv_pw_bf16.hip move 16 bytes per lane along the flattened (sample, position) axis, which
// needs every channel plane to start on a 16-byte boundary.  That holds for every 128-mel configuration,
// but not for the reference's other geometries (mn10_as_mels_40: planes of 5 x 125 and 3 x 63 positions;
// mn10_as_mels_64 with an odd number of frames).  This kernel takes the SAME packed weights and the same
// epilogue contract with plain 4-byte accesses: one wave = 64 consecutive positions of one sample x one
// 16-row m-tile, the k loop on the fp32 VALU (exact fp32 fmaf chain; split bf16 weights are recombined
// hi + lo first; plain-bf16 packs round the activation to bf16 as well, like the MFMA kernel).  Those planes are tiny (<= a few hundred positions), so the kernel is latency-bound and
// not on any measured path; it exists so that the model runs for every input geometry the reference accepts.
#include "eat_common.h"

namespace {

// W[m][k] of the three pack formats (eat_pw_prepack / eat_pw_prepack_bf16)
template <int WMODE>
__device__ __forceinline__ float packed_w(const void* wp, int MT, int mt, int ml, int k) {
  if constexpr (WMODE == 0) {
    return reinterpret_cast<const float*>(wp)[((size_t)(k >> 2) * MT + mt) * 64 + ((k & 3) << 4) + ml];
  } else {
    constexpr int NP2 = WMODE == 2 ? 2 : 1;
    const __bf16* w16 = reinterpret_cast<const __bf16*>(wp);
    const size_t o = ((size_t)((k >> 5) * MT + mt) * NP2) * 512 + (ml + 16 * ((k & 31) >> 3)) * 8 + (k & 7);
    float v = (float)w16[o];
    if constexpr (WMODE == 2) v += (float)w16[o + 512];
    return v;
  }
}

template <int WMODE>
__global__ __launch_bounds__(64) void pw_conv_generic_kernel(
    const float* __restrict__ x, const void* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ pool, int Ci, int Co, int S, int MT, int act, long long wp_bstride_bytes) {
  const int lane = threadIdx.x, mt = blockIdx.y, b = blockIdx.z;
  const int s = blockIdx.x * 64 + lane;
  const bool ok = s < S;
  const int sc = ok ? s : S - 1;
  const char* wb = reinterpret_cast<const char*>(wp) + (size_t)b * wp_bstride_bytes;   // per-sample weights (DyMN) or 0
  const float* xb = x + (size_t)b * Ci * S + sc;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  for (int k = 0; k < Ci; ++k) {
    float xv = xb[(size_t)k * S];
    if (in_scale) xv *= in_scale[(size_t)b * Ci + k];
    if constexpr (WMODE == 1) xv = (float)(__bf16)xv;     // plain-bf16 arithmetic: both operands rounded, as on the MFMA path
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fmaf(packed_w<WMODE>(wb, MT, mt, i, k), xv, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = mt * 16 + i;
    if (m >= Co) break;                                   // block-uniform
    float v = eat::activate_rt(acc[i] + bias[m], act);
    const size_t o = ((size_t)b * Co + m) * S + sc;
    if (res) v += res[o];
    if (ok && y) y[o] = v;
    if (pool) {
      const float ps = eat::wave_sum(ok ? v : 0.0f);
      if (lane == 0) atomicAdd(pool + (size_t)b * Co + m, ps);
    }
  }
}

}  // namespace

namespace eat {

// wmode: 0 = fp32 pack, 1 = bf16 pack, 2 = bf16 hi/lo pack; wp_bstride_bytes != 0: per-sample weights
int pw_conv_generic(const float* x, const void* wp, const float* bias, const float* in_scale, const float* res, float* y,
                    float* pool, int B, int Ci, int Co, int S, int act, int wmode, long long wp_bstride_bytes,
                    hipStream_t s) {
  const int MT = (Co + 15) / 16;
  dim3 grid((S + 63) / 64, MT, B);
  if (wmode == 0)
    hipLaunchKernelGGL(pw_conv_generic_kernel<0>, grid, dim3(64), 0, s, x, wp, bias, in_scale, res, y, pool, Ci, Co, S, MT, act, wp_bstride_bytes);
  else if (wmode == 1)
    hipLaunchKernelGGL(pw_conv_generic_kernel<1>, grid, dim3(64), 0, s, x, wp, bias, in_scale, res, y, pool, Ci, Co, S, MT, act, wp_bstride_bytes);
  else
    hipLaunchKernelGGL(pw_conv_generic_kernel<2>, grid, dim3(64), 0, s, x, wp, bias, in_scale, res, y, pool, Ci, Co, S, MT, act, wp_bstride_bytes);
  return check_launch("eat_pw_conv_fwd (generic plane size)");
}

}  // namespace eat
