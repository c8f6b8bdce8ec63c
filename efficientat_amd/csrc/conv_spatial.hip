// Spatial convolutions of the MobileNetV3 trunk for gfx950: the 3x3/s2 stem and the depthwise
// k x k conv.  Both are HBM-streaming kernels (read the plane once, write it once); eval-mode
// BatchNorm is folded into weights/bias by the caller, the activation and the squeeze
// (global-average-pool partial sums) of SqueezeExcitation are fused into the epilogue.
// Reference call sites: models/mn/model.py:124-133 (stem), models/mn/block_types.py:150-162 (dw),
// models/mn/block_types.py:72-73 (SE mean).
#include <cstdlib>
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
  // range-checked raw buffer loads: a position outside the plane gets the out-of-range offset and reads 0 - no select /
  // branch around the nine loads, they issue back to back (0.21 -> 0.16 ms at B = 256)
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * F * T), 0,
                                                                     (int)(4LL * F * T < 0x7fffffffLL ? 4LL * F * T : 0x7fffffffLL), 0x00020000);
  float in[9];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int fi = 2 * fo + u - 1;
    const bool rok = fi >= 0 && fi < F;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int ti = 2 * to + v - 1;
      const unsigned off = (rok && ti >= 0 && ti < T) ? 4u * (unsigned)(fi * T + ti) : 0x80000000u;
      in[u * 3 + v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xb, (int)off, 0, 0));
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
// One thread = one output column of one (b,c) plane; it walks down the plane keeping the K input
// rows under the filter in registers (a rotating ring, each step only loads the STRIDE new rows,
// one step ahead of their use) with its K*K taps.  Lanes of a wave sit on consecutive columns, so every load / store
// of a step is one coalesced row segment along the time axis; the horizontal overlap between
// neighbouring lanes is served by L1.  No LDS, no barrier, ~70 VGPRs: 7-8 waves per SIMD keep
// enough row segments in flight to stream from HBM.  The activated outputs of a plane are summed
// on the fly for the squeeze of SqueezeExcitation.
//
// DY = true is the DyMN variant (models/dymn/dy_block.py:399-402): the taps are per (b,c) plane (the
// attention-weighted sum of the K=4 kernels, aggregated beforehand), and the epilogue applies
// DyReLU-B  max(a1 v + b1, a2 v + b2)  with per-plane coefficients and the coordinate attention
// sigmoid(g_cf[b,fo,c]) * sigmoid(g_ct[b,to,c]) (gates stored position-major: (B, L, C)).
struct DwDyn {
  const float* coef;    // (B*C, 4): a1, a2, b1, b2
  const float* gate_f;  // (B, Fo, C) pre-sigmoid
  const float* gate_t;  // (B, To, C) pre-sigmoid
  const float* res;     // (B, C, Fo, To) added to the output, or NULL
  int flip;             // read the taps reversed: the stride-1 data gradient is a correlation with flipped taps
  int per_plane_w;      // taps indexed by (b,c) plane instead of channel
  const float* in_a;    // (C) or NULL: the conv input is act_in(in_a[c] * x + in_b[c]) evaluated on load (training:
  const float* in_b;    //   batch-norm + activation of the expand conv fused into the depthwise conv, the activated
  int in_act;           //   tensor is never materialised)
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int K, int STRIDE, int ACT, bool DY>
__global__ __launch_bounds__(256) void dw_conv_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ y, float* __restrict__ pool,
                                                      int n_planes, int C, int F, int T, int Fo, int To,
                                                      int TX, DwDyn dyn) {
  constexpr int P = (K - 1) / 2;
  const int tid = threadIdx.x;
  const int tx = tid % TX, ty = tid / TX;
  const int to = blockIdx.x * TX + tx;
  const int gp = blockIdx.y * (256 / TX) + ty;
  const bool live = gp < n_planes && to < To;
  float psum = 0.0f;
  if (live) {
    const int c = gp % C;
    float wr[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wr[i] = w[(size_t)(dyn.per_plane_w ? gp : c) * K * K + (dyn.flip ? K * K - 1 - i : i)];
    const float bc = bias ? bias[c] : 0.0f;
    float a1 = 1.f, a2 = 0.f, b1 = 0.f, b2 = 0.f, sgt = 1.f;
    const float* gfp = nullptr;
    // coef == NULL and gates == NULL: per-plane taps only (train mode: BN statistics come before DyReLU / CoordAtt);
    // one of them NULL: the `no_dyrelu` / `no_ca` ablations of the block (dy_block.py:353-356)
    const bool dy_relu = DY && dyn.coef != nullptr;
    const bool dy_gate = DY && dyn.gate_f != nullptr;
    if (dy_relu) {
      const float4 cf = *reinterpret_cast<const float4*>(dyn.coef + (size_t)gp * 4);
      a1 = cf.x; a2 = cf.y; b1 = cf.z; b2 = cf.w;
    }
    if (dy_gate) {
      const int b = gp / C;
      sgt = sigmoidf_(dyn.gate_t[((size_t)b * To + to) * C + c]);
      gfp = dyn.gate_f + (size_t)b * Fo * C + c;
    }
    const float* xp = x + (size_t)gp * F * T;
    float* yp = y + (size_t)gp * Fo * To + to;
    const int t0 = to * STRIDE - P;
    bool cok[K];
#pragma unroll
    for (int v = 0; v < K; ++v) cok[v] = (t0 + v >= 0) && (t0 + v < T);

    // Ring of K + STRIDE row slots: while step R multiplies the K rows in slots (u + R*STRIDE) % NSLOT,
    // the STRIDE rows that step R+1 adds are already loading into the other STRIDE slots, so a wave
    // always has the next row segment in flight (one dependent HBM round trip per step otherwise).
    constexpr int NSLOT = K + STRIDE;
    constexpr int PERIOD = (NSLOT % STRIDE == 0) ? NSLOT / STRIDE : NSLOT;   // steps until the ring realigns
    float win[NSLOT][K];
    const bool has_tf = DY && dyn.in_a != nullptr;
    const float ia = has_tf ? dyn.in_a[c] : 1.0f, ib = has_tf ? dyn.in_b[c] : 0.0f;
    auto load_row = [&](int fi, float (&dst)[K]) {
      const bool rok = fi >= 0 && fi < F;              // uniform across the wave
      const float* src = xp + (size_t)(rok ? fi : 0) * T + t0;
      if (has_tf) {                                    // zero padding applies to the TRANSFORMED map
#pragma unroll
        for (int v = 0; v < K; ++v) dst[v] = (rok && cok[v]) ? eat::activate_rt(fmaf(ia, src[v], ib), dyn.in_act) : 0.0f;
      } else {
#pragma unroll
        for (int v = 0; v < K; ++v) dst[v] = (rok && cok[v]) ? src[v] : 0.0f;
      }
    };
#pragma unroll
    for (int u = 0; u < K; ++u) load_row(u - P, win[u]);

    for (int fo0 = 0; fo0 < Fo; fo0 += PERIOD) {
#pragma unroll
      for (int R = 0; R < PERIOD; ++R) {
        const int fo = fo0 + R;
        if (fo < Fo) {
          if (fo + 1 < Fo) {
#pragma unroll
            for (int u = K - STRIDE; u < K; ++u)
              load_row((fo + 1) * STRIDE - P + u, win[(u + (R + 1) * STRIDE) % NSLOT]);
          }
          float acc = bc;
#pragma unroll
          for (int u = 0; u < K; ++u)
#pragma unroll
            for (int v = 0; v < K; ++v) acc = fmaf(wr[u * K + v], win[(u + R * STRIDE) % NSLOT][v], acc);
          float o = eat::activate<ACT>(acc);
          if (dy_relu) o = fmaxf(fmaf(a1, o, b1), fmaf(a2, o, b2));
          if (dy_gate) o *= sigmoidf_(gfp[(size_t)fo * C]) * sgt;
          if (dyn.res) o += dyn.res[(size_t)gp * Fo * To + (size_t)fo * To + to];
          yp[(size_t)fo * To] = o;
          psum += o;
        }
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
              int F, int T, int Fo, int To, int act, const DwDyn* dyn, hipStream_t stream) {
  const int TX = To > 32 ? 64 : 32;
  const int n_planes = B * C;
  const int ppb = 256 / TX;
  dim3 grid((To + TX - 1) / TX, (n_planes + ppb - 1) / ppb);
  if (grid.y > 65535u * 32u) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd: too many planes (%d)", n_planes);
  if (dyn) {
    EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((dw_conv_kernel<K, STRIDE, ACT, true>), grid, dim3(256), 0, stream, x, w, bias,
                                             y, pool, n_planes, C, F, T, Fo, To, TX, *dyn));
  } else {
    EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((dw_conv_kernel<K, STRIDE, ACT, false>), grid, dim3(256), 0, stream, x, w,
                                             bias, y, pool, n_planes, C, F, T, Fo, To, TX, DwDyn{nullptr, nullptr, nullptr, nullptr, 0, 0}));
  }
  return eat::check_launch("eat_dw_conv_fwd");
}

int dispatch_dw(const float* x, const float* w, const float* bias, float* y, float* pool, int B, int C, int F, int T,
                int Fo, int To, int k, int stride, int act, const DwDyn* dyn, hipStream_t s,
                const eat::DwEpi* epi = nullptr) {
  const int p = (k - 1) / 2;
  if (Fo != (F + 2 * p - k) / stride + 1 || To != (T + 2 * p - k) / stride + 1)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd: output %dx%d inconsistent with input %dx%d k=%d s=%d", Fo, To, F, T, k, stride);
  if (!dyn || (!dyn->coef && !dyn->gate_f)) {
    // register-resident kernels (dw_plane.hip: whole small planes, tiles of large ones); 1 = geometry not instantiated
    const int rc = eat::dw_plane_try(x, w, bias, dyn ? dyn->res : nullptr, y, pool, B, C, F, T, Fo, To, k, stride, act,
                                     dyn ? dyn->flip : 0, dyn ? dyn->per_plane_w : 0, dyn ? dyn->in_a : nullptr,
                                     dyn ? dyn->in_b : nullptr, dyn ? dyn->in_act : 0, s, epi);
    if (rc != 1) return rc;
  }
  if (epi) return 1;       // training epilogues exist in the register-resident kernels only: the caller falls back
  if (k == 3 && stride == 1) return launch_dw<3, 1>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, dyn, s);
  if (k == 3 && stride == 2) return launch_dw<3, 2>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, dyn, s);
  if (k == 5 && stride == 1) return launch_dw<5, 1>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, dyn, s);
  if (k == 5 && stride == 2) return launch_dw<5, 2>(x, w, bias, y, pool, B, C, F, T, Fo, To, act, dyn, s);
  return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd: unsupported k=%d stride=%d", k, stride);
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
  return dispatch_dw(x, w, bias, y, pool, B, C, F, T, Fo, To, k, stride, act, nullptr, (hipStream_t)stream);
}

extern "C" int eat_dw_conv_fwd_tf(const float* x, const float* in_a, const float* in_b, int in_act, const float* w,
                                  const float* bias, float* y, int B, int C, int F, int T, int Fo, int To, int k,
                                  int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!in_a || !in_b) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_tf: in_a and in_b are required");
  if (in_act < 0 || in_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_tf: bad in_act %d", in_act);
  const DwDyn dyn{nullptr, nullptr, nullptr, nullptr, 0, 0, in_a, in_b, in_act};
  return dispatch_dw(x, w, bias, y, nullptr, B, C, F, T, Fo, To, k, stride, EAT_ACT_NONE, &dyn, (hipStream_t)stream);
}

// Train-mode depthwise conv (models/mn/block_types.py:150-162 under model.train()): y = conv(act_in(in_a x + in_b)) (in_a
// NULL: plain x) with the per-wave partial sums of y for the BatchNorm that follows (layout [b][2][C][inner] floats;
// *h_inner receives inner, which never exceeds inner_cap = eat_dw_partials_inner(...)).  Geometries without a
// register-resident kernel run the row-ring kernel followed by a one-block-per-plane statistics pass (inner = 1).
static int dw_conv_fwd_stats_impl(int per_plane_w, const float* x, const float* in_a, const float* in_b, int in_act, const float* w,
                                     float* y, float* part, int inner_cap, int* h_inner, int B, int C, int F, int T,
                                     int Fo, int To, int k, int stride, eat_stream_t stream) {
  if ((in_a == nullptr) != (in_b == nullptr)) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats: in_a and in_b come together");
  if (in_act < 0 || in_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats: bad in_act %d", in_act);
  if (!part || !h_inner || inner_cap < eat_dw_partials_inner(F, T, Fo, To, k, stride, 0))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats: partial buffer too small (inner_cap %d)", inner_cap);
  const DwDyn dyn{nullptr, nullptr, nullptr, nullptr, 0, per_plane_w, in_a, in_b, in_act};
  int inner = 1;
  {
    const eat::DwEpi epi{part, nullptr, nullptr, nullptr, 0, nullptr, &inner};
    const int rc = dispatch_dw(x, w, nullptr, y, nullptr, B, C, F, T, Fo, To, k, stride, EAT_ACT_NONE, &dyn, (hipStream_t)stream, &epi);
    if (rc != 1) { *h_inner = inner; return rc; }
  }
  const int rc = dispatch_dw(x, w, nullptr, y, nullptr, B, C, F, T, Fo, To, k, stride, EAT_ACT_NONE, &dyn, (hipStream_t)stream);
  if (rc != 0) return rc;
  *h_inner = 1;
  return eat::bn_stats_partial(y, B, C, Fo * To, part, (hipStream_t)stream);
}

extern "C" int eat_dw_conv_fwd_stats(const float* x, const float* in_a, const float* in_b, int in_act, const float* w,
                                     float* y, float* part, int inner_cap, int* h_inner, int B, int C, int F, int T,
                                     int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  return dw_conv_fwd_stats_impl(0, x, in_a, in_b, in_act, w, y, part, inner_cap, h_inner, B, C, F, T, Fo, To, k, stride, stream);
}

// The same over bf16-stored x and y (act_io.h; the bf16-storage plan of BASELINE configs[2]: the expand conv's output z_e in,
// the depthwise output z_d out, both 16-bit in HBM - x_b16 = 0: x is fp32, the first block's depthwise conv reads the stem
// output; taps, transform coefficients and statistics fp32).  The partial sums are
// those of the ROUNDED outputs - the values the BatchNorm that follows will actually read.  Register-resident kernels only
// (csrc/dw_plane.hip): eat_dw_conv_b16_ok tells whether a geometry is covered.
extern "C" int eat_dw_conv_fwd_stats_b16(const void* x, int x_b16, const float* in_a, const float* in_b, int in_act, const float* w,
                                         void* y, float* part, int inner_cap, int* h_inner, int B, int C, int F, int T,
                                         int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !w || !y) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats_b16: missing operand");
  if ((in_a == nullptr) != (in_b == nullptr)) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats_b16: in_a and in_b come together");
  if (in_act < 0 || in_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats_b16: bad in_act %d", in_act);
  if (!part || !h_inner || inner_cap < eat_dw_partials_inner(F, T, Fo, To, k, stride, 0))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats_b16: partial buffer too small (inner_cap %d)", inner_cap);
  if ((F * T) % 2 != 0 || (Fo * To) % 2 != 0)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats_b16: planes must hold an even number of elements (%d, %d)", F * T, Fo * To);
  int inner = 1;
  const eat::DwEpi epi{part, nullptr, nullptr, nullptr, 0, nullptr, &inner};
  const int rc = eat::dw_plane_try(reinterpret_cast<const float*>(x), w, nullptr, nullptr, reinterpret_cast<float*>(y), nullptr, B,
                                   C, F, T, Fo, To, k, stride, EAT_ACT_NONE, 0, 0, in_a, in_b, in_act, (hipStream_t)stream, &epi,
                                   x_b16 ? 1 : 2);
  if (rc == 1)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_fwd_stats_b16: no register-resident kernel for F=%d T=%d k=%d stride=%d", F, T, k, stride);
  *h_inner = inner;
  return rc;
}

// 1 where eat_dw_conv_fwd_stats_b16 and eat_dw_conv_bwd_bn_g_b16 cover the geometry (host helper: a training plan keeps fp32
// storage for the blocks they do not cover)
extern "C" int eat_dw_conv_b16_ok(int B, int C, int F, int T, int Fo, int To, int k, int stride) {
  if ((F * T) % 2 != 0 || (Fo * To) % 2 != 0 || (long long)B * C > 0x3fffffffLL) return 0;
  if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return 0;
  bool fwd = false;
  if (T > 128 && (long long)F * T < (1 << 28)) fwd = true;                                  // tile kernels
  else if (k == 3 && stride == 1 && F == 8 && T > 32 && T <= 64) fwd = true;               // plane kernels (dw_plane_try)
  else if (k == 5 && stride == 1 && F == 16 && T > 64 && T <= 128) fwd = true;
  else if (k == 5 && stride == 2 && F == 8 && T > 32 && T <= 64) fwd = true;
  else if (k == 3 && stride == 2 && F == 16 && T > 64 && T <= 128) fwd = true;
  else if (k == 5 && stride == 1 && F == 4 && T <= 32) fwd = true;
  return fwd && eat_dw_bwd_merged_ok(B, C, F, T, Fo, To, k, stride) ? 1 : 0;
}

// The same with per-(b,c) taps w_bc (B, C, k*k): DyMN's dynamic depthwise conv in train mode (models/dymn/dy_block.py:
// 103-131 with groups = channels) - the expand BatchNorm + activation evaluated on load, the statistics of depth_norm in
// the epilogue.
extern "C" int eat_dw_conv_dyn_fwd_stats(const float* x, const float* in_a, const float* in_b, int in_act, const float* w_bc,
                                         float* y, float* part, int inner_cap, int* h_inner, int B, int C, int F, int T,
                                         int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  return dw_conv_fwd_stats_impl(1, x, in_a, in_b, in_act, w_bc, y, part, inner_cap, h_inner, B, C, F, T, Fo, To, k, stride,
                                stream);
}

// ... over bf16-stored x and y (the DyMN blocks of the bf16-storage plan; x_b16 = 0: the block without expand conv reads the
// fp32 block input - tile geometries only): eat_dw_conv_fwd_stats_b16 with per-(b,c) taps.
extern "C" int eat_dw_conv_dyn_fwd_stats_b16(const void* x, int x_b16, const float* in_a, const float* in_b, int in_act,
                                             const float* w_bc, void* y, float* part, int inner_cap, int* h_inner, int B, int C,
                                             int F, int T, int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !w_bc || !y) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_fwd_stats_b16: missing operand");
  if ((in_a == nullptr) != (in_b == nullptr)) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_fwd_stats_b16: in_a and in_b come together");
  if (in_act < 0 || in_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_fwd_stats_b16: bad in_act %d", in_act);
  if (!part || !h_inner || inner_cap < eat_dw_partials_inner(F, T, Fo, To, k, stride, 0))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_fwd_stats_b16: partial buffer too small (inner_cap %d)", inner_cap);
  if ((F * T) % 2 != 0 || (Fo * To) % 2 != 0)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_fwd_stats_b16: planes must hold an even number of elements (%d, %d)", F * T, Fo * To);
  int inner = 1;
  const eat::DwEpi epi{part, nullptr, nullptr, nullptr, 0, nullptr, &inner};
  const int rc = eat::dw_plane_try(reinterpret_cast<const float*>(x), w_bc, nullptr, nullptr, reinterpret_cast<float*>(y), nullptr, B,
                                   C, F, T, Fo, To, k, stride, EAT_ACT_NONE, 0, 1, in_a, in_b, in_act, (hipStream_t)stream, &epi,
                                   x_b16 ? 1 : 2);
  if (rc == 1)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_fwd_stats_b16: no register-resident kernel for F=%d T=%d k=%d stride=%d x_b16=%d", F, T, k, stride, x_b16);
  *h_inner = inner;
  return rc;
}

// stride-1 depthwise data gradient = the same sliding-window kernel with the taps read reversed
namespace eat {
int dw_conv_dgrad_s1(const float* dz, const float* w, const float* zero_bias, const float* res, float* dx, int B, int C,
                     int F, int T, int k, int per_plane_w, hipStream_t s, const DwEpi* epi) {
  const DwDyn dyn{nullptr, nullptr, nullptr, res, 1, per_plane_w};
  return dispatch_dw(dz, w, zero_bias, dx, nullptr, B, C, F, T, F, T, k, 1, EAT_ACT_NONE, &dyn, s, epi);
}
}  // namespace eat

extern "C" int eat_dw_conv_dyn_fwd(const float* x, const float* w_bc, const float* bias, const float* coef,
                                   const float* gate_f, const float* gate_t, float* y, int B, int C, int F, int T,
                                   int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  const DwDyn dyn{coef, gate_f, gate_t, nullptr, 0, 1};
  return dispatch_dw(x, w_bc, bias, y, nullptr, B, C, F, T, Fo, To, k, stride, EAT_ACT_NONE, &dyn, (hipStream_t)stream);
}

// Ablations of the dynamic block (models/dymn/dy_block.py:353-356, `no_dyrelu` / `no_ca`): `act` is the plain activation
// that replaces DyReLU-B (applied to the BN output), coef == NULL skips DyReLU-B, gate_f == gate_t == NULL skips the
// coordinate attention.  With everything present and act = none this is eat_dw_conv_dyn_fwd.
extern "C" int eat_dw_conv_dyn_act_fwd(const float* x, const float* w_bc, const float* bias, int act, const float* coef,
                                       const float* gate_f, const float* gate_t, float* y, int B, int C, int F, int T,
                                       int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_act_fwd: bad act %d", act);
  if ((gate_f == nullptr) != (gate_t == nullptr))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_act_fwd: gate_f and gate_t come together");
  const DwDyn dyn{coef, gate_f, gate_t, nullptr, 0, 1};
  return dispatch_dw(x, w_bc, bias, y, nullptr, B, C, F, T, Fo, To, k, stride, act, &dyn, (hipStream_t)stream);
}

// ---- depthwise conv with dilation (models/mn/model.py:244-269 `dilated=True`: the last three blocks run their 5x5
// depthwise conv with dilation 2 and stride 1; torch pads (k-1)/2*dilation).  Not a measured path (no released
// checkpoint uses it): one thread per output element, taps and bias through the scalar cache, reads coalesced along T.
namespace {
__global__ __launch_bounds__(256) void dw_conv_dilated_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              float* __restrict__ pool, int C, int F, int T, int Fo, int To,
                                                              int k, int stride, int dil, int act) {
  const int plane = blockIdx.y, c = plane % C;
  const int pad = (k - 1) / 2 * dil;
  const float* xp = x + (size_t)plane * F * T;
  const float* wc = w + (size_t)c * k * k;
  const float bc = bias[c];
  float ps = 0.0f;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < Fo * To; e += gridDim.x * blockDim.x) {
    const int i = e / To, j = e - i * To;
    float acc = bc;
    for (int u = 0; u < k; ++u) {
      const int fi = i * stride - pad + u * dil;
      if (fi < 0 || fi >= F) continue;
      for (int v = 0; v < k; ++v) {
        const int ti = j * stride - pad + v * dil;
        if (ti >= 0 && ti < T) acc = fmaf(wc[u * k + v], xp[(size_t)fi * T + ti], acc);
      }
    }
    const float o = eat::activate_rt(acc, act);
    y[(size_t)plane * Fo * To + e] = o;
    ps += o;
  }
  if (pool) {
    ps = eat::wave_sum(ps);
    if ((threadIdx.x & 63) == 0) atomicAdd(pool + plane, ps);
  }
}
}  // namespace

extern "C" int eat_dw_conv_dilated_fwd(const float* x, const float* w, const float* bias, float* y, float* pool, int B,
                                       int C, int F, int T, int Fo, int To, int k, int stride, int dilation, int act,
                                       eat_stream_t stream) {
  eat::clear_stale_error();
  if (k < 1 || k > 7 || (k & 1) == 0 || stride < 1 || dilation < 1) return eat::fail(EAT_EINVAL, "eat_dw_conv_dilated_fwd: bad geometry");
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_dilated_fwd: bad act %d", act);
  const int pad = (k - 1) / 2 * dilation;
  if (Fo != (F + 2 * pad - dilation * (k - 1) - 1) / stride + 1 || To != (T + 2 * pad - dilation * (k - 1) - 1) / stride + 1)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dilated_fwd: output %dx%d inconsistent with input %dx%d", Fo, To, F, T);
  int gx = (Fo * To + 255) / 256;
  gx = gx > 32 ? 32 : gx;
  hipLaunchKernelGGL(dw_conv_dilated_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, pool, C, F, T,
                     Fo, To, k, stride, dilation, act);
  return eat::check_launch("eat_dw_conv_dilated_fwd");
}

// Backward of the dilated depthwise conv (training of the `dilated=True` networks; the same generic, unmeasured form):
//   dx[b,c,fi,ti] = sum_{u,v} w[c,u,v] dz[b,c,i,j] over the outputs with i*stride - pad + u*dil = fi (same for j / ti);
//   dw[c,u,v]     = sum_{b,i,j} dz[b,c,i,j] x[b,c,i*stride - pad + u*dil, j*stride - pad + v*dil]  (atomics, dw zeroed).
namespace {
__global__ __launch_bounds__(256) void dw_dilated_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                               float* __restrict__ dx, int C, int F, int T, int Fo, int To,
                                                               int k, int stride, int dil) {
  const int plane = blockIdx.y, c = plane % C;
  const int pad = (k - 1) / 2 * dil;
  const float* g = dz + (size_t)plane * Fo * To;
  const float* wc = w + (size_t)c * k * k;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * T; e += gridDim.x * blockDim.x) {
    const int fi = e / T, ti = e - fi * T;
    float acc = 0.0f;
    for (int u = 0; u < k; ++u) {
      const int ii = fi + pad - u * dil;
      if (ii < 0 || ii % stride != 0 || ii / stride >= Fo) continue;
      for (int v = 0; v < k; ++v) {
        const int jj = ti + pad - v * dil;
        if (jj >= 0 && jj % stride == 0 && jj / stride < To)
          acc = fmaf(wc[u * k + v], g[(size_t)(ii / stride) * To + jj / stride], acc);
      }
    }
    dx[(size_t)plane * F * T + e] = acc;
  }
}

// one block per (tap, channel): the 256 threads walk the channel's output positions of every sample
__global__ __launch_bounds__(256) void dw_dilated_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                               float* __restrict__ dw, int B, int C, int F, int T, int Fo,
                                                               int To, int k, int stride, int dil) {
  __shared__ float s_red[4];
  const int tap = blockIdx.x, c = blockIdx.y;
  const int u = tap / k, v = tap - u * k;
  const int pad = (k - 1) / 2 * dil;
  float acc = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* g = dz + ((size_t)b * C + c) * Fo * To;
    const float* xp = x + ((size_t)b * C + c) * F * T;
    for (int e = threadIdx.x; e < Fo * To; e += 256) {
      const int i = e / To, j = e - i * To;
      const int fi = i * stride - pad + u * dil, ti = j * stride - pad + v * dil;
      if (fi >= 0 && fi < F && ti >= 0 && ti < T) acc = fmaf(g[e], xp[(size_t)fi * T + ti], acc);
    }
  }
  acc = eat::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dw[(size_t)c * k * k + tap] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
}  // namespace

static int dilated_geometry_ok(int F, int T, int Fo, int To, int k, int stride, int dilation) {
  if (k < 1 || k > 7 || (k & 1) == 0 || stride < 1 || dilation < 1) return 0;
  const int pad = (k - 1) / 2 * dilation;
  return Fo == (F + 2 * pad - dilation * (k - 1) - 1) / stride + 1 && To == (T + 2 * pad - dilation * (k - 1) - 1) / stride + 1;
}

extern "C" int eat_dw_conv_dilated_dgrad(const float* dz, const float* w, float* dx, int B, int C, int F, int T, int Fo, int To,
                                         int k, int stride, int dilation, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dilated_geometry_ok(F, T, Fo, To, k, stride, dilation)) return eat::fail(EAT_EINVAL, "eat_dw_conv_dilated_dgrad: bad geometry");
  int gx = (F * T + 255) / 256;
  gx = gx > 32 ? 32 : gx;
  hipLaunchKernelGGL(dw_dilated_dgrad_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, dz, w, dx, C, F, T, Fo, To, k,
                     stride, dilation);
  return eat::check_launch("eat_dw_conv_dilated_dgrad");
}

extern "C" int eat_dw_conv_dilated_wgrad(const float* dz, const float* x, float* dw, int B, int C, int F, int T, int Fo, int To,
                                         int k, int stride, int dilation, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dilated_geometry_ok(F, T, Fo, To, k, stride, dilation)) return eat::fail(EAT_EINVAL, "eat_dw_conv_dilated_wgrad: bad geometry");
  hipLaunchKernelGGL(dw_dilated_wgrad_kernel, dim3(k * k, C), dim3(256), 0, (hipStream_t)stream, dz, x, dw, B, C, F, T, Fo, To, k,
                     stride, dilation);
  return eat::check_launch("eat_dw_conv_dilated_wgrad");
}
