// Fused  expand 1x1 conv + BN + act  ->  depthwise k x k conv + BN + act (+ SE squeeze sums)  for gfx950.
//   reference: the first two ConvNormActivation stages of InvertedResidual
//   (models/mn/block_types.py:138-162) and the SE mean (:72-73).
//
// Why: the expanded tensor (C_exp = 3-6 x C_in channels at the block's INPUT resolution) is the
// largest activation of every block; written by the expand conv and read back by the depthwise
// conv it is ~48 % of the whole network's HBM traffic.  Here it never leaves the CU:
//
//   block = (sample b, 16 expanded channels, TFo x TTo output tile)
//   1. the C_in x (TFi x TTi) input patch (tile + halo) streams through LDS in 16-row K chunks
//      (plain coalesced loads: patch rows are short unaligned segments, not LDS-DMA material)
//   2. fp32 MFMA (16x16x4) accumulates the 16 x P expanded patch in registers; bias + activation,
//      positions outside the image forced to 0 (the depthwise conv zero-pads the ACTIVATED map)
//   3. the patch is written to LDS (re-using the chunk buffer) and the depthwise conv runs on it:
//      each thread owns one channel (taps in registers) and a strip of outputs
//   4. outputs go to HBM, plane sums to the SE accumulator.
//
// The input patch is re-read once per 16-channel chunk (from L2: all chunks of a tile sit on one
// XCD), and the halo is recomputed ((TFi*TTi)/(TFo*TTo*s*s) - 1, 13-45 % extra expand flops), so
// this is used for the early, bandwidth-bound blocks (C_in <= 40); the late blocks (8x63 / 4x32
// planes, C_in >= 80) are MFMA-bound and keep the separate kernels.
#include "eat_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kKCh = 16;       // K rows per chunk
constexpr int kCC = 16;        // expanded channels per block
constexpr int kMaxGPW = 3;     // 64-position groups per wave (patch <= 768 positions)

template <int K, int STRIDE, int ACT>
__global__ __launch_bounds__(256, 2) void fused_expand_dw_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias_e,
    const float* __restrict__ wd, const float* __restrict__ bias_d, float* __restrict__ y,
    float* __restrict__ pool, int Cin, int Cexp, int F, int T, int Fo, int To, int MT, int TFo, int TTo,
    int tiles_t, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int P_ = (K - 1) / 2;
  const int TFi = (TFo - 1) * STRIDE + K, TTi = (TTo - 1) * STRIDE + K;
  const int P = TFi * TTi;
  const int Ppad = (P + 63) & ~63;
  const int NG = Ppad >> 6;
  float* Xs = smem;                       // [kKCh][Ppad]  (later: the expanded patch [kCC][Ppad])
  float* As = smem + kKCh * Ppad;         // [kKCh/4][64] A fragments of this chunk
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // XCD-aware: the channel chunks of one tile have ids congruent mod 8 (same XCD, x patch shared in L2)
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
  const int mt = jj % MT, tile = (jj / MT) * 8 + xcd;
  const int b = blockIdx.y;
  if (tile >= n_tiles) return;
  const int tf = tile / tiles_t, tt = tile - tf * tiles_t;
  const int fo0 = tf * TFo, to0 = tt * TTo;
  const int fi0 = fo0 * STRIDE - P_, ti0 = to0 * STRIDE - P_;
  const size_t plane_in = (size_t)F * T;
  const float* xb = x + (size_t)b * Cin * plane_in;

  // loader role: up to 3 patch positions per thread, same positions for every k row
  int off[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int pos = tid + 256 * q;
    off[q] = -1;
    if (pos < P) {
      const int pi = pos / TTi, pj = pos - pi * TTi;
      const int fi = fi0 + pi, ti = ti0 + pj;
      if (fi >= 0 && fi < F && ti >= 0 && ti < T) off[q] = fi * T + ti;
    }
  }
  // MFMA role: wave w owns position groups g = w + 4q; lane owns positions 64 g + 4 (lane&15) .. +3
  const int kq = lane >> 4;
  f32x4 acc[kMaxGPW][4];
#pragma unroll
  for (int q = 0; q < kMaxGPW; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < Cin; k0 += kKCh) {
    const int klen = (Cin - k0) < kKCh ? (Cin - k0) : kKCh;
    __syncthreads();                                   // previous chunk fully consumed
    // all loads of the chunk are issued before the first LDS store (one HBM/L2 round trip per chunk,
    // not one per k row)
    float stage[kKCh][3];
#pragma unroll
    for (int r = 0; r < kKCh; ++r) {
      const float* src = xb + (size_t)(k0 + (r < klen ? r : 0)) * plane_in;
#pragma unroll
      for (int q = 0; q < 3; ++q) stage[r][q] = (r < klen && off[q] >= 0) ? src[off[q]] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < kKCh; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int pos = tid + 256 * q;
        if (r < klen && pos < Ppad) Xs[r * Ppad + pos] = stage[r][q];
      }
    if (tid < (klen >> 2) * 64) As[tid] = wp[((size_t)((k0 >> 2) + (tid >> 6)) * MT + mt) * 64 + (tid & 63)];
    __syncthreads();
    const int ksteps = klen >> 2;
    for (int ks = 0; ks < ksteps; ++ks) {
      const float a = As[ks * 64 + lane];
#pragma unroll
      for (int q = 0; q < kMaxGPW; ++q) {
        const int g = wv + 4 * q;
        if (g < NG) {
          const float4 xv = *reinterpret_cast<const float4*>(Xs + (ks * 4 + kq) * Ppad + 64 * g + 4 * (lane & 15));
          acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.x, acc[q][0], 0, 0, 0);
          acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.y, acc[q][1], 0, 0, 0);
          acc[q][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.z, acc[q][2], 0, 0, 0);
          acc[q][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.w, acc[q][3], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();                                     // everyone is done reading Xs: it becomes the patch
  float* Es = smem;                                    // [kCC][Ppad]
#pragma unroll
  for (int q = 0; q < kMaxGPW; ++q) {
    const int g = wv + 4 * q;
    if (g < NG) {
      const int pos0 = 64 * g + 4 * (lane & 15);
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pos = pos0 + j;
        const int pi = pos / TTi, pj = pos - pi * TTi;
        const int fi = fi0 + pi, ti = ti0 + pj;
        ok[j] = pos < P && fi >= 0 && fi < F && ti >= 0 && ti < T;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = kq * 4 + r;                      // C/D layout: row = kq*4 + r, col = lane & 15
        const int cg = mt * 16 + ch;
        const float be = cg < Cexp ? bias_e[cg] : 0.0f;
        float4 v;
        v.x = ok[0] ? eat::activate<ACT>(acc[q][0][r] + be) : 0.0f;
        v.y = ok[1] ? eat::activate<ACT>(acc[q][1][r] + be) : 0.0f;
        v.z = ok[2] ? eat::activate<ACT>(acc[q][2][r] + be) : 0.0f;
        v.w = ok[3] ? eat::activate<ACT>(acc[q][3][r] + be) : 0.0f;
        *reinterpret_cast<float4*>(Es + ch * Ppad + pos0) = v;
      }
    }
  }
  __syncthreads();

  // depthwise stage: each (channel, tile column) pair is walked down the tile rows by one thread with
  // the K x K window in registers (K*STRIDE LDS reads per output instead of K*K); lanes sit on
  // consecutive columns of one channel, so LDS reads and the global stores are row segments.
  for (int pr = tid; pr < kCC * TTo; pr += 256) {
    const int ch = pr / TTo, tl = pr - ch * TTo;
    const int cg = mt * 16 + ch, to = to0 + tl;
    float psum = 0.0f;
    if (cg < Cexp && to < To) {
      float wr[K * K];
#pragma unroll
      for (int i = 0; i < K * K; ++i) wr[i] = wd[(size_t)cg * K * K + i];
      const float bd = bias_d[cg];
      const float* ep = Es + ch * Ppad + tl * STRIDE;
      float* yp = y + ((size_t)b * Cexp + cg) * Fo * To + to;
      float win[K][K];
#pragma unroll
      for (int u = 0; u < K - STRIDE; ++u)
#pragma unroll
        for (int v = 0; v < K; ++v) win[u][v] = ep[u * TTi + v];
      for (int fl0 = 0; fl0 < TFo; fl0 += K) {
#pragma unroll
        for (int R = 0; R < K; ++R) {                 // K steps = one full rotation of the window slots
          const int fl = fl0 + R;
          if (fl < TFo && fo0 + fl < Fo) {
#pragma unroll
            for (int u = K - STRIDE; u < K; ++u)
#pragma unroll
              for (int v = 0; v < K; ++v) win[(u + R * STRIDE) % K][v] = ep[(fl * STRIDE + u) * TTi + v];
            float sacc = bd;
#pragma unroll
            for (int u = 0; u < K; ++u)
#pragma unroll
              for (int v = 0; v < K; ++v) sacc = fmaf(wr[u * K + v], win[(u + R * STRIDE) % K][v], sacc);
            const float ov = eat::activate<ACT>(sacc);
            yp[(size_t)(fo0 + fl) * To] = ov;
            psum += ov;
          }
        }
      }
    }
    if (pool) {     // TTo (16 or 32) consecutive lanes share the channel: one atomic per channel per block
      for (int o = TTo >> 1; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
      if (tl == 0 && cg < Cexp) atomicAdd(pool + (size_t)b * Cexp + cg, psum);
    }
  }
}

template <int K, int STRIDE>
int launch_fused(const float* x, const float* wp, const float* bias_e, const float* wd, const float* bias_d, float* y,
                 float* pool, int B, int Cin, int Cexp, int F, int T, int Fo, int To, int act, hipStream_t s) {
  // tile: 32 output columns (16 for the wide stride-2 / 5x5 patches), rows chosen so the patch fits 768 positions
  int TTo = 32, TFo = 8;
  auto patch = [&](int tfo, int tto) { return ((tfo - 1) * STRIDE + K) * ((tto - 1) * STRIDE + K); };
  if (TFo > Fo) TFo = Fo;
  while (patch(TFo, TTo) > 64 * 4 * kMaxGPW && TFo > 1) TFo >>= 1;
  if (patch(TFo, TTo) > 64 * 4 * kMaxGPW) TTo = 16;
  if (patch(TFo, TTo) > 64 * 4 * kMaxGPW) return eat::fail(EAT_EINVAL, "eat_fused_expand_dw_fwd: patch too large");
  const int P = patch(TFo, TTo), Ppad = (P + 63) & ~63;
  const int tiles_t = (To + TTo - 1) / TTo, tiles_f = (Fo + TFo - 1) / TFo;
  const int n_tiles = tiles_t * tiles_f;
  const int MT = (Cexp + 15) / 16;
  const size_t smem = ((size_t)kKCh * Ppad + 64 * (kKCh / 4)) * sizeof(float);
  const int tiles8 = (n_tiles + 7) / 8 * 8;
  dim3 grid(tiles8 * MT, B);
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((fused_expand_dw_kernel<K, STRIDE, ACT>), grid, dim3(256), smem, s, x, wp, bias_e,
                                           wd, bias_d, y, pool, Cin, Cexp, F, T, Fo, To, MT, TFo, TTo, tiles_t, n_tiles));
  return eat::check_launch("eat_fused_expand_dw_fwd");
}

}  // namespace

extern "C" int eat_fused_expand_dw_fwd(const float* x, const float* wp_e, const float* bias_e, const float* w_d,
                                       const float* bias_d, float* y, float* pool, int B, int Cin, int Cexp, int F,
                                       int T, int Fo, int To, int k, int stride, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (Cin % 4 != 0) return eat::fail(EAT_EINVAL, "eat_fused_expand_dw_fwd: Cin=%d must be a multiple of 4", Cin);
  if (act != EAT_ACT_RELU && act != EAT_ACT_HSWISH) return eat::fail(EAT_EINVAL, "eat_fused_expand_dw_fwd: act must be relu/hswish");
  const int p = (k - 1) / 2;
  if (Fo != (F + 2 * p - k) / stride + 1 || To != (T + 2 * p - k) / stride + 1)
    return eat::fail(EAT_EINVAL, "eat_fused_expand_dw_fwd: output %dx%d inconsistent with input %dx%d", Fo, To, F, T);
  hipStream_t s = (hipStream_t)stream;
  if (k == 3 && stride == 1) return launch_fused<3, 1>(x, wp_e, bias_e, w_d, bias_d, y, pool, B, Cin, Cexp, F, T, Fo, To, act, s);
  if (k == 3 && stride == 2) return launch_fused<3, 2>(x, wp_e, bias_e, w_d, bias_d, y, pool, B, Cin, Cexp, F, T, Fo, To, act, s);
  if (k == 5 && stride == 1) return launch_fused<5, 1>(x, wp_e, bias_e, w_d, bias_d, y, pool, B, Cin, Cexp, F, T, Fo, To, act, s);
  if (k == 5 && stride == 2) return launch_fused<5, 2>(x, wp_e, bias_e, w_d, bias_d, y, pool, B, Cin, Cexp, F, T, Fo, To, act, s);
  return eat::fail(EAT_EINVAL, "eat_fused_expand_dw_fwd: unsupported k=%d stride=%d", k, stride);
}
