// DyMN-specific kernels for gfx950 (models/dymn/dy_block.py):
//   ctx_pool      - the two average pools of ContextGen (:236-237): x (B,C,F,T) -> one position-major
//                   sequence (B, F+T, C) holding the T-means (rows 0..F-1) and the F-means (rows F..F+T-1)
//   dyn_aggregate - Equation 6 / :111-117: per-sample kernel = sum_k attention[b,k] * bank[k] (generic,
//                   used for the depthwise taps, with the eval BatchNorm scale folded per channel)
//   dyn_pw_pack   - the same aggregation for a 1x1 dynamic conv, written directly in the MFMA A-fragment
//                   order eat_pw_conv_dyn_fwd consumes (one packed matrix per sample)
// The reference materialises (B*Cout, Cin/g, k, k) aggregated weights with a batched matmul and runs a
// grouped conv with groups*B; here the aggregation is one streaming kernel and the conv is the same
// MFMA / sliding-window kernel as the static network, reading per-sample weights.
#include "eat_common.h"

namespace {

// one block per (b,c) plane; wave w takes rows w, w+4, ...
__global__ __launch_bounds__(256) void ctx_pool_kernel(const float* __restrict__ x, float* __restrict__ seq,
                                                       int C, int F, int T) {
  extern __shared__ float s_col[];                       // [4][T]
  const int plane = blockIdx.x, b = plane / C, c = plane % C;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* xp = x + (size_t)plane * F * T;
  float* out = seq + (size_t)b * (F + T) * C + c;
  float* mycol = s_col + wv * T;
  for (int t = lane; t < T; t += 64) mycol[t] = 0.0f;
  for (int f = wv; f < F; f += 4) {
    float rs = 0.0f;
    for (int t = lane; t < T; t += 64) {
      const float v = xp[(size_t)f * T + t];
      rs += v;
      mycol[t] += v;                                       // lane-private slot: no race
    }
    rs = eat::wave_sum(rs);
    if (lane == 0) out[(size_t)f * C] = rs / (float)T;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256)
    out[(size_t)(F + t) * C] = (s_col[t] + s_col[T + t] + s_col[2 * T + t] + s_col[3 * T + t]) / (float)F;
}

// out[b, n] = gscale[n / group] * sum_k att[b,k] * bank[k, n]
__global__ __launch_bounds__(256) void dyn_aggregate_kernel(const float* __restrict__ bank, const float* __restrict__ att,
                                                            const float* __restrict__ gscale, float* __restrict__ out,
                                                            int K, int N, int group) {
  const int b = blockIdx.y;
  const float* a = att + (size_t)b * K;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) acc = fmaf(a[k], bank[(size_t)k * N + n], acc);
    out[(size_t)b * N + n] = gscale ? acc * gscale[n / group] : acc;
  }
}

// wp[b][(ks*MT + mt)*64 + lane] = rs[m] * sum_k att[b,k] * bank[k][m*Ci + kc],  m = mt*16 + (lane&15), kc = ks*4 + (lane>>4)
__global__ __launch_bounds__(256) void dyn_pw_pack_kernel(const float* __restrict__ bank, const float* __restrict__ att,
                                                          const float* __restrict__ row_scale, float* __restrict__ wp,
                                                          int K, int Co, int Ci, int MT) {
  const int b = blockIdx.y;
  const int total = (Ci / 4) * MT * 64;
  const float* a = att + (size_t)b * K;
  const size_t N = (size_t)Co * Ci;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, mt = (i >> 6) % MT, ks = (i >> 6) / MT;
    const int m = mt * 16 + (lane & 15), kc = ks * 4 + (lane >> 4);
    float v = 0.0f;
    if (m < Co) {
      const size_t n = (size_t)m * Ci + kc;
      for (int k = 0; k < K; ++k) v = fmaf(a[k], bank[(size_t)k * N + n], v);
      if (row_scale) v *= row_scale[m];
    }
    wp[(size_t)b * total + i] = v;
  }
}

}  // namespace

extern "C" int eat_ctx_pool(const float* x, float* seq, int B, int C, int F, int T, eat_stream_t stream) {
  eat::clear_stale_error();
  const size_t smem = (size_t)4 * T * sizeof(float);
  if (smem > 64 * 1024) return eat::fail(EAT_EINVAL, "eat_ctx_pool: T=%d too wide", T);
  hipLaunchKernelGGL(ctx_pool_kernel, dim3(B * C), dim3(256), smem, (hipStream_t)stream, x, seq, C, F, T);
  return eat::check_launch("eat_ctx_pool");
}

extern "C" int eat_dyn_aggregate(const float* bank, const float* att, const float* gscale, float* out, int B, int K,
                                 int N, int group, eat_stream_t stream) {
  eat::clear_stale_error();
  if (group < 1) return eat::fail(EAT_EINVAL, "eat_dyn_aggregate: group must be >= 1");
  int gx = (N + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(dyn_aggregate_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, bank, att, gscale, out, K, N,
                     group);
  return eat::check_launch("eat_dyn_aggregate");
}

extern "C" int eat_dyn_pw_pack(const float* bank, const float* att, const float* row_scale, float* wp, int B, int K,
                               int Co, int Ci, eat_stream_t stream) {
  eat::clear_stale_error();
  if (Ci % 4 != 0) return eat::fail(EAT_EINVAL, "eat_dyn_pw_pack: Ci=%d must be a multiple of 4", Ci);
  const int MT = (Co + 15) / 16;
  const int total = (Ci / 4) * MT * 64;
  int gx = (total + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(dyn_pw_pack_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, bank, att, row_scale, wp, K, Co,
                     Ci, MT);
  return eat::check_launch("eat_dyn_pw_pack");
}
