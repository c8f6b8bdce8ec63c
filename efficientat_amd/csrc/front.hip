// Network front in one kernel for gfx950:  stem 3x3/s2 conv + BN + Hardswish  ->  first inverted-residual
// block (no expand, no SE): depthwise 3x3 + BN + act -> project 1x1 + BN -> + residual.
//   reference: models/mn/model.py:124-133 (stem) and the first row of the block table
//   (models/mn/model.py:275, block_types.py:150-181: expanded == input channels, stride 1).
//
// Why: at the stem resolution (64 x 500 for a 10 s clip) every tensor is 2 MB per clip and the separate
// kernels move 12.5 MB per clip (stem 0.5 + 2, depthwise 2 + 2, project 2 + 2 + 2 residual) through HBM
// for ~0.4 GFLOP/clip.  Fused, only the log-mel patch is read and the block output written: 2.5 MB per clip.
//
//   block = 4 waves on (sample b, 8 x 32 output tile), C = 16 channels
//   A. the 21 x 69 log-mel patch under the tile (+ halo of the two stacked 3x3 convs) -> LDS, zero outside
//   B. stem: thread = patch position, all 16 channels, channel pairs on packed fp32 FMAs, weights via
//      scalar loads; Hardswish; positions outside the stem plane forced to 0 (the depthwise conv zero-pads
//      its input); 16 x (10 x 34) stem tile -> LDS
//   C. depthwise: thread = (channel, tile column) walking down the 8 rows with the 10 x 3 strip in
//      registers; 16 x 256 outputs -> LDS
//   D. project: fp32 MFMA 16x16x4 (K = 16 channels; accumulators start at the bias), + residual from the
//      stem tile still in LDS, float4 stores.
#include "eat_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int kC = 16;
constexpr int kTF = 8, kTT = 32;               // output tile
constexpr int kSF = kTF + 2, kST = kTT + 2;    // stem tile (halo 1)
constexpr int kSP = kSF * kST;                 // 340 stem positions
constexpr int kSPad = 352;                     // row stride of Ss (floats)
constexpr int kMF = 2 * kSF + 1, kMT = 2 * kST + 1;   // 21 x 69 log-mel patch
constexpr int kMTPad = 72;

template <int ACT>
__global__ __launch_bounds__(256) void front_kernel(const float* __restrict__ x, const float* __restrict__ ws,
                                                    const float* __restrict__ bs, const float* __restrict__ wd,
                                                    const float* __restrict__ bd, const float* __restrict__ wpp,
                                                    const float* __restrict__ bp, float* __restrict__ y, int F, int T,
                                                    int Fo, int To, int tiles_t) {
  __shared__ __attribute__((aligned(16))) float Ms[kMF * kMTPad];
  __shared__ __attribute__((aligned(16))) float Ss[kC * kSPad];
  __shared__ __attribute__((aligned(16))) float Ds[kC * kTF * kTT];
  __shared__ float s_wd[kC * 9 + kC];          // depthwise taps + bias: staged once per block (phase C read them with
                                               // 18 per-thread global loads, a dependent L2 round trip per pass)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kq = lane >> 4;
  const int tile = blockIdx.x, b = blockIdx.y;
  const int tf = tile / tiles_t, tt = tile - tf * tiles_t;
  const int fo0 = tf * kTF, to0 = tt * kTT;
  const int sf0 = fo0 - 1, st0 = to0 - 1;                 // first stem row / column of the tile
  const int mf0 = 2 * sf0 - 1, mt0 = 2 * st0 - 1;         // first log-mel row / column
  const float* xb = x + (size_t)b * F * T;

  // project A fragments (4 k-steps) and its bias row: in flight during phases A-C
  float ap[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ap[ks] = wpp[ks * 64 + lane];
  f32x4 bpv;
#pragma unroll
  for (int r = 0; r < 4; ++r) bpv[r] = bp[kq * 4 + r];

  if (tid < kC * 9) s_wd[tid] = wd[tid];
  else if (tid < kC * 9 + kC) s_wd[tid] = bd[tid - kC * 9];
  // ---- A. log-mel patch
  for (int e = tid; e < kMF * kMTPad; e += 256) {
    const int i = e / kMTPad, j = e - i * kMTPad;
    const int fi = mf0 + i, ti = mt0 + j;
    Ms[e] = (j < kMT && fi >= 0 && fi < F && ti >= 0 && ti < T) ? xb[(size_t)fi * T + ti] : 0.0f;
  }
  __syncthreads();

  // ---- B. stem on the 10 x 34 tile
  for (int p = tid; p < kSP; p += 256) {
    const int pi = p / kST, pj = p - pi * kST;
    const int sf = sf0 + pi, st = st0 + pj;
    const bool inside = sf >= 0 && sf < Fo && st >= 0 && st < To;
    float m[9];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) m[u * 3 + v] = Ms[(2 * pi + u) * kMTPad + 2 * pj + v];
#pragma unroll
    for (int c = 0; c < kC; c += 2) {
      f32x2 acc = {bs[c], bs[c + 1]};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f32x2 w2 = {ws[c * 9 + t], ws[(c + 1) * 9 + t]};
        const f32x2 m2 = {m[t], m[t]};
        acc = w2 * m2 + acc;
      }
      Ss[c * kSPad + p] = inside ? eat::activate<EAT_ACT_HSWISH>(acc[0]) : 0.0f;
      Ss[(c + 1) * kSPad + p] = inside ? eat::activate<EAT_ACT_HSWISH>(acc[1]) : 0.0f;
    }
  }
  __syncthreads();

  // ---- C. depthwise 3x3 / stride 1 on the stem tile
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int ch = ps * 8 + (tid >> 5), tl = tid & 31;
    const float* sp = Ss + ch * kSPad + tl;
    float col[kSF][3];
#pragma unroll
    for (int u = 0; u < kSF; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) col[u][v] = sp[u * kST + v];
    float wr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wr[i] = s_wd[ch * 9 + i];
    const float bdc = s_wd[kC * 9 + ch];
#pragma unroll
    for (int fl = 0; fl < kTF; ++fl) {
      float s = bdc;
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) s = fmaf(wr[u * 3 + v], col[fl + u][v], s);
      Ds[ch * (kTF * kTT) + fl * kTT + tl] = eat::activate<ACT>(s);
    }
  }
  __syncthreads();

  // ---- D. project 16 -> 16 on the matrix cores; wave w owns outputs 64 w .. 64 w + 63 (two tile rows)
  f32x4 acc[4] = {bpv, bpv, bpv, bpv};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const float4 dv = *reinterpret_cast<const float4*>(Ds + (ks * 4 + kq) * (kTF * kTT) + 64 * wv + 4 * (lane & 15));
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks], dv.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks], dv.y, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks], dv.z, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks], dv.w, acc[3], 0, 0, 0);
  }
  const int p0 = 64 * wv + 4 * (lane & 15);
  const int fl = p0 >> 5, tl = p0 & 31;
  const int fo = fo0 + fl, to = to0 + tl;
  if (fo < Fo && to < To) {
    const bool vec = ((To & 3) == 0) && (to + 3 < To);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = kq * 4 + r;
      const float* rp = Ss + ch * kSPad + (fl + 1) * kST + tl + 1;      // residual = the block's input
      float* yp = y + (((size_t)b * kC + ch) * Fo + fo) * To + to;
      const float v0 = acc[0][r] + rp[0], v1 = acc[1][r] + rp[1], v2 = acc[2][r] + rp[2], v3 = acc[3][r] + rp[3];
      if (vec) {
        *reinterpret_cast<float4*>(yp) = make_float4(v0, v1, v2, v3);
      } else {
        yp[0] = v0;
        if (to + 1 < To) yp[1] = v1;
        if (to + 2 < To) yp[2] = v2;
        if (to + 3 < To) yp[3] = v3;
      }
    }
  }
}

}  // namespace

extern "C" int eat_front_fwd(const float* x, const float* w_s, const float* bias_s, const float* w_d,
                             const float* bias_d, const float* wp_p, const float* bias_p, float* y, int B, int C, int F,
                             int T, int Fo, int To, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (C != kC) return eat::fail(EAT_EINVAL, "eat_front_fwd: C=%d unsupported (the fused front is built for 16 channels)", C);
  if (Fo != (F - 1) / 2 + 1 || To != (T - 1) / 2 + 1)
    return eat::fail(EAT_EINVAL, "eat_front_fwd: output %dx%d does not match input %dx%d", Fo, To, F, T);
  if (act != EAT_ACT_RELU && act != EAT_ACT_HSWISH) return eat::fail(EAT_EINVAL, "eat_front_fwd: act must be relu/hswish");
  {   // round 2: register-resident kernel (irb.hip, FRONT mode); this LDS-staged one is the fallback / A-B reference
    const int rc = eat::front_try(x, w_s, bias_s, w_d, bias_d, wp_p, bias_p, y, B, C, F, T, Fo, To, act, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  const int tiles_t = (To + kTT - 1) / kTT, tiles_f = (Fo + kTF - 1) / kTF;
  dim3 grid(tiles_t * tiles_f, B);
  hipStream_t s = (hipStream_t)stream;
  if (act == EAT_ACT_RELU)
    hipLaunchKernelGGL((front_kernel<EAT_ACT_RELU>), grid, dim3(256), 0, s, x, w_s, bias_s, w_d, bias_d, wp_p, bias_p, y, F, T,
                       Fo, To, tiles_t);
  else
    hipLaunchKernelGGL((front_kernel<EAT_ACT_HSWISH>), grid, dim3(256), 0, s, x, w_s, bias_s, w_d, bias_d, wp_p, bias_p, y, F,
                       T, Fo, To, tiles_t);
  return eat::check_launch("eat_front_fwd");
}
