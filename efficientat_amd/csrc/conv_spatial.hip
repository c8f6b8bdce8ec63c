// Spatial convolutions of the MobileNetV3 trunk for gfx950: the 3x3/s2 stem and the depthwise
// k x k conv.  Both are HBM-streaming kernels (read the plane once, write it once); eval-mode
// BatchNorm is folded into weights/bias by the caller, the activation and the squeeze
// (global-average-pool partial sums) of SqueezeExcitation are fused into the epilogue.
// Reference call sites: models/mn/model.py:124-133 (stem), models/mn/block_types.py:150-162 (dw),
// models/mn/block_types.py:72-73 (SE mean).
#include "eat_common.h"

namespace {

// ---------------------------------------------------------------------------------- stem
// One thread = one output pixel, all C output channels; lanes walk the time axis so every
// per-channel store is a coalesced 256 B row segment.
template <int ACT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, int C, int F, int T,
                                                        int Fo, int To) {
  const int to = blockIdx.x * 64 + (threadIdx.x & 63);
  const int fo = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (to >= To || fo >= Fo) return;
  const float* xb = x + (size_t)b * F * T;
  float in[9];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int fi = 2 * fo + u - 1;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int ti = 2 * to + v - 1;
      in[u * 3 + v] = (fi >= 0 && fi < F && ti >= 0 && ti < T) ? xb[(size_t)fi * T + ti] : 0.0f;
    }
  }
  float* yb = y + ((size_t)b * C * Fo + fo) * To + to;
  const size_t plane = (size_t)Fo * To;
  for (int c = 0; c < C; ++c) {
    const float* wc = w + c * 9;   // wave-uniform -> scalar loads
    float acc = bias[c];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc = fmaf(wc[i], in[i], acc);
    yb[c * plane] = eat::activate<ACT>(acc);
  }
}

// ----------------------------------------------------------------------------- depthwise
// A block stages NC consecutive (b,c) planes x the input rows of TR output rows (+halo, zero
// padded) in LDS; thread (tx, ty) owns plane ty % NC, output rows ty / NC + i*RS and columns
// tx + j*TX, with its k*k filter taps in registers.  blockDim.x = TX * NC * RS = 256.
template <int K, int STRIDE, int ACT>
__global__ __launch_bounds__(256) void dw_conv_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ y, float* __restrict__ pool,
                                                      int n_planes, int C, int F, int T, int Fo, int To,
                                                      int TX, int NC, int RS, int TR) {
  extern __shared__ __attribute__((aligned(16))) float s_in[];
  constexpr int P = (K - 1) / 2;
  const int IR = (TR - 1) * STRIDE + K;      // staged input rows per plane
  const int W = T + 2 * P;                   // staged row width (zero halo)
  const int tid = threadIdx.x;
  const int tx = tid % TX, ty = tid / TX;
  const int plane0 = blockIdx.y * NC;
  const int fo0 = blockIdx.x * TR;
  const int fi0 = fo0 * STRIDE - P;

  // stage: thread rows walk (plane, input row); lanes walk the time axis (coalesced)
  const int row_groups = 256 / TX;
  for (int rr = ty; rr < NC * IR; rr += row_groups) {
    const int pl = rr / IR, ir = rr - pl * IR;
    const int fi = fi0 + ir;
    const int gp = plane0 + pl;
    const bool row_ok = (gp < n_planes) && fi >= 0 && fi < F;
    const float* src = x + ((size_t)gp * F + fi) * T;
    float* dst = s_in + (size_t)rr * W;
    for (int t = tx; t < W; t += TX) {
      const int ti = t - P;
      dst[t] = (row_ok && ti >= 0 && ti < T) ? src[ti] : 0.0f;
    }
  }
  __syncthreads();

  const int pl = ty % NC, rsub = ty / NC;
  const int gp = plane0 + pl;
  float psum = 0.0f;
  if (gp < n_planes) {
    const int c = gp % C;
    float wr[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wr[i] = w[c * K * K + i];
    const float bc = bias[c];
    const float* sp = s_in + (size_t)pl * IR * W;
    float* yp = y + (size_t)gp * Fo * To;
    for (int r = rsub; r < TR; r += RS) {
      const int fo = fo0 + r;
      if (fo >= Fo) break;
      const float* srow = sp + (size_t)(r * STRIDE) * W;
      for (int to = tx; to < To; to += TX) {
        const float* s0 = srow + to * STRIDE;
        float acc = bc;
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
          for (int v = 0; v < K; ++v) acc = fmaf(wr[u * K + v], s0[u * W + v], acc);
        const float o = eat::activate<ACT>(acc);
        yp[(size_t)fo * To + to] = o;
        psum += o;
      }
    }
  }
  if (pool != nullptr) {
    // lanes sharing a plane inside a wave: min(TX, 64) consecutive lanes
    const int span = TX < 64 ? TX : 64;
    for (int o = span >> 1; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
    if ((tid & (span - 1)) == 0 && gp < n_planes) atomicAdd(pool + gp, psum);
  }
}

template <int K, int STRIDE>
int launch_dw(const float* x, const float* w, const float* bias, float* y, float* pool, int B, int C,
              int F, int T, int Fo, int To, int act, hipStream_t stream) {
  int TX = 32;
  while (TX < To && TX < 256) TX <<= 1;
  const int rest = 256 / TX;
  int NC, RS, TR;
  if (Fo <= 8) { NC = rest; RS = 1; TR = Fo; }
  else { NC = 1; RS = rest; TR = 8 * RS; if (TR > 16) TR = 16; if (TR < 8) TR = 8; }
  const int n_planes = B * C;
  auto smem_of = [&](int nc, int tr) { return (size_t)nc * ((tr - 1) * STRIDE + K) * (T + K - 1) * sizeof(float); };
  while (smem_of(NC, TR) > 48 * 1024 && NC > 1) { NC >>= 1; RS <<= 1; }
  while (smem_of(NC, TR) > 48 * 1024 && TR > 1) TR >>= 1;
  if (smem_of(NC, TR) > 64 * 1024) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd: row of %d floats too wide for LDS tile", T);
  dim3 grid((Fo + TR - 1) / TR, (n_planes + NC - 1) / NC);
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((dw_conv_kernel<K, STRIDE, ACT>), grid, dim3(256), smem_of(NC, TR),
                                           stream, x, w, bias, y, pool, n_planes, C, F, T, Fo, To, TX, NC, RS, TR));
  return eat::check_launch("eat_dw_conv_fwd");
}

}  // namespace

extern "C" int eat_stem_conv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C,
                                 int F, int T, int Fo, int To, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (Fo != (F - 1) / 2 + 1 || To != (T - 1) / 2 + 1)
    return eat::fail(EAT_EINVAL, "eat_stem_conv_fwd: output %dx%d does not match input %dx%d", Fo, To, F, T);
  dim3 grid((To + 63) / 64, (Fo + 3) / 4, B);
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((stem_conv_kernel<ACT>), grid, dim3(256), 0, (hipStream_t)stream, x, w,
                                           bias, y, C, F, T, Fo, To));
  return eat::check_launch("eat_stem_conv_fwd");
}

extern "C" int eat_dw_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* pool,
                               int B, int C, int F, int T, int Fo, int To, int k, int stride, int act,
                               eat_stream_t stream) {
  eat::clear_stale_error();
  const int p = (k - 1) / 2;
  if (Fo != (F + 2 * p - k) / stride + 1 || To != (T + 2 * p - k) / stride + 1)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd: output %dx%d inconsistent with input %dx%d k=%d s=%d", Fo, To, F, T, k, stride);
  hipStream_t s = (hipStream_t)stream;
  if (k == 3 && stride == 1) return launch_dw<3, 1>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, s);
  if (k == 3 && stride == 2) return launch_dw<3, 2>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, s);
  if (k == 5 && stride == 1) return launch_dw<5, 1>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, s);
  if (k == 5 && stride == 2) return launch_dw<5, 2>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, s);
  return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd: unsupported k=%d stride=%d", k, stride);
}
