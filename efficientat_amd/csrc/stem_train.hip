// Stem of the training step without its pre-activation tensor (round 3).
//   reference: models/mn/model.py:124-133 - Conv2d(1, C, 3, stride 2, padding 1, bias=False) -> BatchNorm2d (train) ->
//   Hardswish on the (B, 1, F, T) log-mel; autograd over it.
//
// The stem output z0 (B, C, F/2, T/2) is 4x the log-mel per 16 channels and was passed over 10 times per step (written,
// read for the statistics, read by BN + act; backward: reduce 2 reads, apply 2 reads + 1 write, weight gradient 1 read +
// the window).  The conv is linear in the 9-tap patch p of the log-mel, z0 = W p, so the algebra of train_fuse.hip applies
// with the patch as the "input channels":
//   statistics: sum z = W sp, sum z^2 = rowsum((W G9) .* W),  G9 = sum p p^T (9 x 9), sp = sum p   - a pass over the log-mel;
//   forward   : y0 = hswish(a (W p) + b) straight from the log-mel (eat_stem_conv_fwd with folded weights): z0 never exists;
//   backward  : g = dy0 * hswish'(a (W p) + b) with W p recomputed from the window that the weight gradient holds in
//               registers anyway, Gx = sum g p^T, S1 = sum g  -> eat_expand_bwd_coef gives dW, dgamma, dbeta.
// One pass over dy0 replaces seven.
#include "eat_common.h"

namespace {

constexpr int kNG = 54;          // 45 upper-triangle products + 9 sums

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Raw buffer loads (as dw_plane.hip): a lane whose offset is kOOB reads 0 (hardware range check) - the zero padding of the
// conv without a branch or a select around the load, so the nine loads of a window issue back to back.
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
  const int n = bytes < 0x7fffffffLL ? (int)bytes : 0x7fffffff;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
// 3x3 window of the log-mel plane `r` under stem output (i, j)
__device__ __forceinline__ void load_window(__amdgpu_buffer_rsrc_t r, int i, int j, int F, int T, float (&xw)[9]) {
  unsigned vc[3];
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    const int ti = 2 * j + v - 1;
    vc[v] = (ti >= 0 && ti < T) ? 4u * (unsigned)ti : kOOB;
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int fi = 2 * i + u - 1;
    const bool rok = fi >= 0 && fi < F;                              // wave-uniform
    const unsigned so = rok ? 4u * (unsigned)(fi * T) : 0u;
#pragma unroll
    for (int v = 0; v < 3; ++v)
      xw[u * 3 + v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(rok ? vc[v] : kOOB), (int)so, 0));
  }
}

// per-block partials of G9 (upper triangle, row-major) and sp; part[(b * gridDim.x + blockIdx.x) * 54 + e]
__global__ __launch_bounds__(256) void stem_gram_kernel(const float* __restrict__ x, float* __restrict__ part, int F, int T,
                                                        int Fo, int To, int rows_per_block) {
  __shared__ float s_red[4][kNG];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * rows_per_block;
  const int i1 = (i0 + rows_per_block) < Fo ? (i0 + rows_per_block) : Fo;
  const __amdgpu_buffer_rsrc_t xb = make_rsrc(x + (size_t)b * F * T, 4LL * F * T);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float acc[kNG];
#pragma unroll
  for (int e = 0; e < kNG; ++e) acc[e] = 0.0f;
  for (int i = i0; i < i1; ++i)
    for (int j = threadIdx.x; j < To; j += 256) {
      float xw[9];
      load_window(xb, i, j, F, T, xw);
#pragma unroll
      for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int v = u; v < 9; ++v) {
          constexpr int dummy = 0; (void)dummy;
          const int e = u * 9 - (u * (u - 1)) / 2 + (v - u);        // compile-time after unrolling
          acc[e] = fmaf(xw[u], xw[v], acc[e]);
        }
#pragma unroll
      for (int u = 0; u < 9; ++u) acc[45 + u] += xw[u];
    }
#pragma unroll
  for (int e = 0; e < kNG; ++e) {
    const float v = eat::wave_sum(acc[e]);
    if (lane == 0) s_red[wv][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNG) {
    const int e = threadIdx.x;
    part[((size_t)b * gridDim.x + blockIdx.x) * kNG + e] = (s_red[0][e] + s_red[1][e]) + (s_red[2][e] + s_red[3][e]);
  }
}

// fixed-order fp64 reduction of the partials -> Tm = W G9 (C x 9), sp (9).  One block of 1024 threads.
__global__ __launch_bounds__(1024) void stem_gram_finalize_kernel(const float* __restrict__ part, int nblk,
                                                                  const float* __restrict__ W, int C,
                                                                  float* __restrict__ Tm, float* __restrict__ sp) {
  __shared__ double s_g[kNG];
  __shared__ double s_G[81];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int e = wv; e < kNG; e += nw) {
    double s = 0.0;
    for (int k = lane; k < nblk; k += 64) s += (double)part[(size_t)k * kNG + e];
    s = wave_sum_d(s);
    if (lane == 0) s_g[e] = s;
  }
  __syncthreads();
  if (threadIdx.x < 81) {
    int u = threadIdx.x / 9, v = threadIdx.x % 9;
    if (u > v) { const int t = u; u = v; v = t; }
    s_G[threadIdx.x] = s_g[u * 9 - (u * (u - 1)) / 2 + (v - u)];        // row u of the upper triangle starts at u*9 - u(u-1)/2
  }
  if (threadIdx.x < 9) sp[threadIdx.x] = (float)s_g[45 + threadIdx.x];
  __syncthreads();
  for (int e = threadIdx.x; e < C * 9; e += blockDim.x) {
    const int c = e / 9, k = e - c * 9;
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) t += (double)W[c * 9 + j] * s_G[j * 9 + k];
    Tm[e] = (float)t;
  }
}

// Block partials of  Gx[c][t] = sum g p_t,  s1[c] = sum g  with  g = (dy + dy2) * act'(a_c (W_c . p) + b_c):
// part[block][C][10].  A thread owns output columns and keeps
// 10 sums per channel; the window is loaded once per position.  The 9 taps + (a, b) of the channels sit in LDS and are
// re-read per position as broadcast 16-byte reads (176 values do not fit the scalar registers, and per-position scalar
// loads expose their latency: 0.87 ms; the memory clobber keeps the compiler from hoisting them into 192 VGPRs).
// Round 5: channel groups of 4 (122 VGPRs, 4 waves per SIMD; groups of 8 took 232) and a one-deep software pipeline over
// the (row, column-pass) iterations; same-box microbenchmark at B = 128, C = 64: 741 -> 625 us, B = 256, C = 16: 408 -> 344 us
// (bit-identical sums: a thread adds the same values in the same order).
template <int CG>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dy2,
                                                       const float* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ a,
                                                       const float* __restrict__ bb, int act, float* __restrict__ part,
                                                       int C, int F, int T, int Fo, int To, int rows_per_block) {
  __shared__ float s_red[4][CG * 10];
  __shared__ __attribute__((aligned(16))) float s_w[CG][12];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * rows_per_block;
  const int i1 = (i0 + rows_per_block) < Fo ? (i0 + rows_per_block) : Fo;
  const __amdgpu_buffer_rsrc_t xb = make_rsrc(x + (size_t)b * F * T, 4LL * F * T);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool is_hs = act == EAT_ACT_HSWISH, is_re = act == EAT_ACT_RELU;
  for (int c0 = 0; c0 < C; c0 += CG) {
    __syncthreads();
    for (int e = threadIdx.x; e < CG * 12; e += 256) {
      const int c = e / 12, t = e - c * 12;
      float v = 0.0f;
      if (c0 + c < C) v = t < 9 ? w[(c0 + c) * 9 + t] : (t == 9 ? a[c0 + c] : (t == 10 ? bb[c0 + c] : 0.0f));
      s_w[c][t] = v;
    }
    __syncthreads();
    float acc[CG][10];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int t = 0; t < 10; ++t) acc[c][t] = 0.0f;
    // software pipeline over the block's (row, column-pass) iterations: the window and the CG gradient values of the NEXT
    // iteration are in flight while this one is evaluated.  Range-checked buffer loads (no divergent branches): a channel
    // past C or a column past To reads 0.
    const int plane = Fo * To;
    const int cg = (C - c0) < CG ? (C - c0) : CG;
    const __amdgpu_buffer_rsrc_t gb = make_rsrc(dy + ((size_t)b * C + c0) * plane, 4LL * cg * plane);
    const __amdgpu_buffer_rsrc_t gb2 = make_rsrc(dy2 ? dy2 + ((size_t)b * C + c0) * plane : dy, dy2 ? 4LL * cg * plane : 0LL);
    const int KT = (To + 255) >> 8;
    float xn[9], dn[CG], dn2[CG];
    int ii = i0, kk = 0;
    auto fetch = [&]() {
      const int j = threadIdx.x + (kk << 8);
      load_window(xb, ii, j, F, T, xn);
      const unsigned vo = j < To ? 4u * (unsigned)(ii * To + j) : kOOB;
#pragma unroll
      for (int c = 0; c < CG; ++c) dn[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gb, (int)vo, 4 * c * plane, 0));
#pragma unroll
      for (int c = 0; c < CG; ++c) dn2[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gb2, (int)vo, 4 * c * plane, 0));
      if (++kk == KT) { kk = 0; ++ii; }
    };
    const int n_it = (i1 - i0) * KT;
    if (n_it > 0) fetch();
    for (int it = 0; it < n_it; ++it) {
      float xw[9], dv[CG];
#pragma unroll
      for (int t = 0; t < 9; ++t) xw[t] = xn[t];
#pragma unroll
      for (int c = 0; c < CG; ++c) dv[c] = dn[c] + dn2[c];
      if (it + 1 < n_it) fetch();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const float4 w0 = *reinterpret_cast<const float4*>(&s_w[c][0]);
        const float4 w1 = *reinterpret_cast<const float4*>(&s_w[c][4]);
        const float4 w2 = *reinterpret_cast<const float4*>(&s_w[c][8]);
        float z = w0.x * xw[0];
        z = fmaf(w0.y, xw[1], z); z = fmaf(w0.z, xw[2], z); z = fmaf(w0.w, xw[3], z);
        z = fmaf(w1.x, xw[4], z); z = fmaf(w1.y, xw[5], z); z = fmaf(w1.z, xw[6], z); z = fmaf(w1.w, xw[7], z);
        z = fmaf(w2.x, xw[8], z);
        const float u = fmaf(w2.y, z, w2.z);
        const float m_in = (u >= -3.0f && u <= 3.0f) ? 1.0f : 0.0f, m_hi = u > 3.0f ? 1.0f : 0.0f;
        const float dhs = fmaf(fmaf(u, 1.0f / 3.0f, 0.5f), m_in, m_hi);
        const float dre = u > 0.0f ? 1.0f : 0.0f;
        const float d = is_hs ? dhs : (is_re ? dre : 1.0f);
        const float g = dv[c] * d;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(g, xw[t], acc[c][t]);
        acc[c][9] += g;
      }
    }
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        const float v = eat::wave_sum(acc[c][t]);
        if (lane == 0) s_red[wv][c * 10 + t] = v;
      }
    __syncthreads();
    // plain stores of the block's sums (4096 blocks x 160 atomics on five cache lines were the whole cost of the first
    // version: ~0.9 ns per atomic); stem_bwd_finalize_kernel adds the blocks in a fixed order
    float* pb = part + ((size_t)b * gridDim.x + blockIdx.x) * C * 10;
    for (int e = threadIdx.x; e < CG * 10; e += 256) {
      const int c = e / 10;
      if (c0 + c < C) pb[(size_t)c0 * 10 + e] = (s_red[0][e] + s_red[1][e]) + (s_red[2][e] + s_red[3][e]);
    }
  }
}

// one block per channel: gx[c][0..9) and s1[c] = sum over the nblk block partials (fp64, fixed order)
__global__ __launch_bounds__(256) void stem_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C,
                                                                float* __restrict__ gx, float* __restrict__ s1) {
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int t = wv; t < 10; t += 4) {
    double s = 0.0;
    for (int k = lane; k < nblk; k += 64) s += (double)part[((size_t)k * C + c) * 10 + t];
    s = wave_sum_d(s);
    if (lane == 0) {
      if (t < 9) gx[c * 9 + t] = (float)s;
      else s1[c] = (float)s;
    }
  }
}

}  // namespace

extern "C" int eat_stem_gram_blocks(int B, int Fo) {
  const int rpb = 8;
  return B * ((Fo + rpb - 1) / rpb);
}

extern "C" int eat_stem_gram(const float* x, const float* W, float* part, float* Tm, float* sp, int B, int C, int F, int T,
                             eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || F < 1 || T < 1) return eat::fail(EAT_EINVAL, "eat_stem_gram: bad shape");
  const int Fo = (F - 1) / 2 + 1, To = (T - 1) / 2 + 1, rpb = 8;
  const dim3 grid((unsigned)((Fo + rpb - 1) / rpb), (unsigned)B);
  hipLaunchKernelGGL(stem_gram_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, part, F, T, Fo, To, rpb);
  hipLaunchKernelGGL(stem_gram_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, (int)(grid.x * grid.y), W,
                     C, Tm, sp);
  return eat::check_launch("eat_stem_gram");
}

extern "C" int eat_stem_bwd_blocks(int B, int Fo) {
  const int rpb = 8;
  return B * ((Fo + rpb - 1) / rpb);
}

extern "C" int eat_stem_bwd(const float* dy, const float* dy2, const float* x, const float* W, const float* a, const float* b,
                            int act, float* part, float* gx, float* s1, int B, int C, int F, int T, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || F < 1 || T < 1) return eat::fail(EAT_EINVAL, "eat_stem_bwd: bad shape");
  const int Fo = (F - 1) / 2 + 1, To = (T - 1) / 2 + 1, rpb = 8;
  const dim3 grid((unsigned)((Fo + rpb - 1) / rpb), (unsigned)B);
  hipLaunchKernelGGL(stem_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, dy2, x, W, a, b, act, part, C, F, T, Fo,
                     To, rpb);
  hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, part,
                     (int)(grid.x * grid.y), C, gx, s1);
  return eat::check_launch("eat_stem_bwd");
}
