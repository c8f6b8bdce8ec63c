// Training-step kernels, round 3: BatchNorm statistics and BatchNorm-backward reductions WITHOUT their own passes over
// the expanded tensors (reference: nn.BatchNorm2d(eps=1e-3, momentum=0.01) in train mode, models/mn/model.py:114-115,
// autograd over models/mn/block_types.py:138-181).
//
//  (1) per-wave partial sums written by the producing kernels (dw_plane.hip epilogues) are reduced here, per channel, in
//      fp64 - no atomics, no zero-filled accumulators, bit-reproducible;
//  (2) the expand 1x1 conv z = W x has its batch statistics from the Gram matrix of its (3-6x narrower) input:
//      sum z = W sx, sum z^2 = diag(W G W^T) with sx = sum x, G = sum x x^T - z is never read for statistics;
//  (3) its BatchNorm backward is linear in xhat = (W x - mu) invstd, so with g = dy * act'(.) (written by the depthwise
//      data-gradient kernel's epilogue) and Gx = sum g x^T:
//        dgamma = invstd * (rowsum(W .* Gx) - mu S1),  dbeta = S1 = sum g,  m1 = S1 / N,  m2 = dgamma / N
//        dW  = diag(a) [Gx - m1 sx^T - diag(m2 invstd)(W G - mu sx^T)]
//        dx  = (W^T diag(a)) g + M x + c0,   M = -W^T diag(a m2 invstd) W,   c0 = W^T (a (m2 invstd mu - m1))
//      i.e. the tensor dz = a (g - m1 - xhat m2) is never formed: two 1x1 convs (one over the narrow input) replace the
//      reduce pass, the apply pass and its re-reads.
#include <cstdlib>
#include "eat_common.h"
#include "act_io.h"

namespace {

__device__ __forceinline__ float act_deriv(float u, int act) {
  if (act == EAT_ACT_RELU) return u > 0.0f ? 1.0f : 0.0f;
  if (act == EAT_ACT_HSWISH) return u < -3.0f ? 0.0f : (u <= 3.0f ? fmaf(u, 1.0f / 3.0f, 0.5f) : 1.0f);
  return 1.0f;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of up to 3 doubles (blockDim.x = 256); result valid in every thread
__device__ __forceinline__ void block_sum_d(double& a, double& b, double& c, double* s_red) {
  a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();                              // s_red may still be read from a previous call
  if (lane == 0) { s_red[wv] = a; s_red[4 + wv] = b; s_red[8 + wv] = c; }
  __syncthreads();
  double ta = 0.0, tb = 0.0, tc = 0.0;
  for (int i = 0; i < nw; ++i) { ta += s_red[i]; tb += s_red[4 + i]; tc += s_red[8 + i]; }
  a = ta; b = tb; c = tc;
}

// ---- generic producers of the partials (any geometry): one block per (b,c) plane, inner = 1 ------------------------
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, int C, int S,
                                                               float* __restrict__ part) {
  __shared__ float s_red[16];
  const int plane = blockIdx.x, c = plane % C, b = plane / C;
  const float* p = z + (size_t)plane * S;
  float s1 = 0.f, s2 = 0.f;
  if ((S & 3) == 0) {
#pragma unroll 4
    for (int i = threadIdx.x * 4; i < S; i += blockDim.x * 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + i);
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (int i = threadIdx.x; i < S; i += blockDim.x) { const float v = p[i]; s1 += v; s2 += v * v; }
  }
  s1 = eat::wave_sum(s1); s2 = eat::wave_sum(s2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { s_red[wv] = s1; s_red[8 + wv] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { ta += s_red[i]; tb += s_red[8 + i]; }
    part[((size_t)b * 2 + 0) * C + c] = ta;
    part[((size_t)b * 2 + 1) * C + c] = tb;
  }
}

// g = dy * act'(a_c z + b_c) (may be written in place of dy); gpart[b*C + c] = sum_s g
__global__ __launch_bounds__(256) void act_grad_sum_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           int act, float* g, float* __restrict__ gpart, int C, int S) {
  __shared__ float s_red[8];
  const int plane = blockIdx.x, c = plane % C;
  const float av = a[c], bv = b[c];
  const size_t base = (size_t)plane * S;
  float s1 = 0.f;
  if ((S & 3) == 0) {
#pragma unroll 4
    for (int i = threadIdx.x * 4; i < S; i += blockDim.x * 4) {
      const float4 d = *reinterpret_cast<const float4*>(dy + base + i);
      const float4 v = *reinterpret_cast<const float4*>(z + base + i);
      const float4 o = make_float4(d.x * act_deriv(fmaf(av, v.x, bv), act), d.y * act_deriv(fmaf(av, v.y, bv), act),
                                   d.z * act_deriv(fmaf(av, v.z, bv), act), d.w * act_deriv(fmaf(av, v.w, bv), act));
      *reinterpret_cast<float4*>(g + base + i) = o;
      s1 += (o.x + o.y) + (o.z + o.w);
    }
  } else {
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
      const float o = dy[base + i] * act_deriv(fmaf(av, z[base + i], bv), act);
      g[base + i] = o;
      s1 += o;
    }
  }
  s1 = eat::wave_sum(s1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) s_red[wv] = s1;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
    gpart[plane] = t;
  }
}

// ---- squeeze-excitation blocks: ONE pass over (d, z) for both the gate gradient and the BatchNorm-backward sums -----
// Forward: y = act(a z + b), xs = y * s[b,c] (models/mn/block_types.py:72-83).  Backward with d = d(xs):
//   d s[b,c] = sum_s d y                      -> through the gate MLP -> gadd[b,c] (the squeeze mean's gradient)
//   g = (d * s + gadd) * act'(u),  BatchNorm backward needs sum g and sum g (z - mu) per channel.
// gadd depends on d s, so round 2 read (d, z) twice (plane_dot, then the reduce pass).  s and gadd are per-PLANE constants:
//   sum g = s * P1 + gadd * P2,   sum g (z - mu) = s * P3 + gadd * P4
// with P0 = sum d y, P1 = sum d act', P2 = sum act', P3 = sum d act' (z - mu), P4 = sum act' (z - mu) per plane - all
// five taken here in one pass; se_bn_bwd_combine_kernel finishes per channel once the gate gradients are known.
template <int ACT, typename ZT = float>
__global__ __launch_bounds__(256) void se_bn_bwd_partials_kernel(const ZT* __restrict__ d, const ZT* __restrict__ z,
                                                                 const float* __restrict__ a, const float* __restrict__ b,
                                                                 const float* __restrict__ mean, float* __restrict__ P,
                                                                 int C, int S, int n_planes) {
  __shared__ float s_red[5][4];
  const int plane = blockIdx.x, c = plane % C;
  const float av = a[c], bv = b[c], mu = mean[c];
  const size_t base = (size_t)plane * S;
  float p[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  auto acc1 = [&](float dv, float zv) {
    const float u = fmaf(av, zv, bv);
    const float da = act_deriv(u, ACT), y = eat::activate<ACT>(u), zc = zv - mu;
    p[0] = fmaf(dv, y, p[0]);
    p[1] = fmaf(dv, da, p[1]);
    p[2] += da;
    p[3] = fmaf(dv * da, zc, p[3]);
    p[4] = fmaf(da, zc, p[4]);
  };
  if ((S & 3) == 0) {
#pragma unroll 2
    for (int i = threadIdx.x * 4; i < S; i += blockDim.x * 4) {
      const float4 dv = eat::Io<ZT>::load4(d + base + i);
      const float4 zv = eat::Io<ZT>::load4(z + base + i);
      acc1(dv.x, zv.x); acc1(dv.y, zv.y); acc1(dv.z, zv.z); acc1(dv.w, zv.w);
    }
  } else {
    for (int i = threadIdx.x; i < S; i += blockDim.x) acc1(eat::Io<ZT>::load1(d + base + i), eat::Io<ZT>::load1(z + base + i));
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const float t = eat::wave_sum(p[q]);
    if (lane == 0) s_red[q][wv] = t;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += s_red[threadIdx.x][i];
    P[(size_t)threadIdx.x * n_planes + plane] = t;
  }
}

// Small planes (S <= 512, S % 4 == 0: the 8 x 63 and 4 x 32 planes of the late SE blocks): one block per plane is a 64-thread
// block that reads 1 - 4 KB - block scheduling, not bandwidth, set the pace (2.6 TB/s at S = 128).  Here a WAVE owns LPP-lane
// groups, one plane per group (LPP = 32 for S <= 128: two planes per wave), walks its planes with a grid stride and
// reduces inside the group with xor shuffles: no LDS, no barrier.
template <int ACT, int LPP, typename ZT = float>
__global__ __launch_bounds__(256) void se_bn_bwd_partials_small_kernel(const ZT* __restrict__ d, const ZT* __restrict__ z,
                                                                       const float* __restrict__ a, const float* __restrict__ b,
                                                                       const float* __restrict__ mean, float* __restrict__ P,
                                                                       int C, int S, int n_planes) {
  constexpr int GPW = 64 / LPP;                                      // plane groups per wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int grp = lane / LPP, gl = lane % LPP;
  const int stride = gridDim.x * 4 * GPW;
  for (int plane = (blockIdx.x * 4 + wv) * GPW + grp; plane < n_planes; plane += stride) {
    const int c = plane % C;
    const float av = a[c], bv = b[c], mu = mean[c];
    const size_t base = (size_t)plane * S;
    float p[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = gl * 4; i < S; i += LPP * 4) {
      const float4 dv = eat::Io<ZT>::load4(d + base + i);
      const float4 zv = eat::Io<ZT>::load4(z + base + i);
      const float dd[4] = {dv.x, dv.y, dv.z, dv.w}, zz[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = fmaf(av, zz[e], bv);
        const float da = act_deriv(u, ACT), y = eat::activate<ACT>(u), zc = zz[e] - mu;
        p[0] = fmaf(dd[e], y, p[0]);
        p[1] = fmaf(dd[e], da, p[1]);
        p[2] += da;
        p[3] = fmaf(dd[e] * da, zc, p[3]);
        p[4] = fmaf(da, zc, p[4]);
      }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
#pragma unroll
      for (int o = LPP / 2; o > 0; o >>= 1) p[q] += __shfl_xor(p[q], o, 64);
    }
    if (gl < 5) P[(size_t)gl * n_planes + plane] = gl == 0 ? p[0] : gl == 1 ? p[1] : gl == 2 ? p[2] : gl == 3 ? p[3] : p[4];
  }
}

// sums[c] = sum_b (s P1 + gadd P2), sums[C + c] = invstd[c] * sum_b (s P3 + gadd P4)   (fp64, one block per channel)
__global__ __launch_bounds__(64) void se_bn_bwd_combine_kernel(const float* __restrict__ P, const float* __restrict__ gs,
                                                               const float* __restrict__ ga,
                                                               const float* __restrict__ invstd, int B, int C,
                                                               double* __restrict__ sums) {
  const int c = blockIdx.x;
  const size_t np = (size_t)B * C;
  double s1 = 0.0, s2 = 0.0;
  for (int bb = threadIdx.x; bb < B; bb += 64) {
    const size_t pl = (size_t)bb * C + c;
    const double s = (double)gs[pl], g = (double)ga[pl];
    s1 += s * (double)P[1 * np + pl] + g * (double)P[2 * np + pl];
    s2 += s * (double)P[3 * np + pl] + g * (double)P[4 * np + pl];
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (threadIdx.x == 0) {
    sums[c] = s1;
    sums[C + c] = s2 * (double)invstd[c];
  }
}

// ---- BatchNorm finalize from partials [outer][2][C][inner]: one block per channel ------------------------------------
// Round 5: with many partial rows (the 1x1 epilogue leaves one row of 2 C sums per 256-column tile: 32 000 rows for the
// 16-channel first project conv at B = 256) one block per channel walks its column of the [outer][2 C inner] matrix one
// cache line per element - 92 us for 4 MB.  bn_partials_rowgroups_kernel sums row groups with coalesced reads first (fp64,
// fixed order: thread (lane, column) adds rows lane, lane + RL, ...; the RL lanes are added in index order) and leaves
// [groups][2 C inner] doubles for the finalize kernel: 2 launches of ~5 us.
constexpr int kFinGroupsMax = 64;
__host__ __device__ __forceinline__ int fin_col_threads(int W) { int cw = 32; while (cw < W && cw < 256) cw <<= 1; return cw; }
__host__ __device__ __forceinline__ int fin_groups(int outer, int C, int inner) {
  if ((long long)outer * inner < 2048) return 0;                    // the direct kernel is as fast there
  const int W = 2 * C * inner, rl = 256 / fin_col_threads(W);
  int g = outer / (rl * 8);
  return g < 1 ? 1 : (g > kFinGroupsMax ? kFinGroupsMax : g);
}
__global__ __launch_bounds__(256) void bn_partials_rowgroups_kernel(const float* __restrict__ part, int outer, int W,
                                                                    double* __restrict__ ws) {
  __shared__ double s_acc[256];
  const int cw = fin_col_threads(W), rl = 256 / cw;
  const int col = blockIdx.x * cw + (threadIdx.x % cw), lane = threadIdx.x / cw;
  const int G = gridDim.y, per = (outer + G - 1) / G;
  const int r0 = blockIdx.y * per, r1 = (r0 + per) < outer ? (r0 + per) : outer;
  double acc = 0.0;
  if (col < W) {
    int r = r0 + lane;
    for (; r + 3 * rl < r1; r += 4 * rl) {
      const float v0 = part[(size_t)r * W + col], v1 = part[(size_t)(r + rl) * W + col];
      const float v2 = part[(size_t)(r + 2 * rl) * W + col], v3 = part[(size_t)(r + 3 * rl) * W + col];
      acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
    }
    for (; r < r1; r += rl) acc += (double)part[(size_t)r * W + col];
  }
  s_acc[threadIdx.x] = acc;
  __syncthreads();
  if (lane == 0 && col < W) {
    double t = 0.0;
    for (int l = 0; l < rl; ++l) t += s_acc[l * cw + (threadIdx.x % cw)];
    ws[(size_t)blockIdx.y * W + col] = t;
  }
}

template <typename PT>
__global__ __launch_bounds__(256) void bn_finalize_partials_kernel(
    const PT* __restrict__ part, int outer, int C, int inner, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
    float eps, double n, float* __restrict__ a, float* __restrict__ b, float* __restrict__ mean,
    float* __restrict__ invstd) {
  __shared__ double s_red[12];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0, dummy = 0.0;
  // thread t owns the entries (o, i) with o * inner + i = t (mod 256); inner <= 256: i and o advance without a division,
  // and the loads of four entries are issued before their sums (the entries of one channel are 2 C inner floats apart:
  // every load is its own cache line, the loop was a chain of dependent-latency round trips - 9.6 us per launch, 31 launches)
  const size_t ostride = (size_t)2 * C * inner, koff = (size_t)C * inner;
  int o = threadIdx.x / inner, i = threadIdx.x - o * inner;
  const int od = 256 / inner, id = 256 - od * inner;
  auto next = [&]() { i += id; o += od; if (i >= inner) { i -= inner; ++o; } };
  if (inner <= 256) {
    while (o < outer) {
      PT v1[4], v2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = o < outer;
        const PT* pp = part + (size_t)(ok ? o : 0) * ostride + (size_t)c * inner + i;
        v1[q] = ok ? pp[0] : (PT)0;
        v2[q] = ok ? pp[koff] : (PT)0;
        next();
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { s1 += (double)v1[q]; s2 += (double)v2[q]; }
    }
  } else {
    const int total = outer * inner;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int oo = e / inner, ii = e - oo * inner;
      s1 += (double)part[(((size_t)oo * 2 + 0) * C + c) * inner + ii];
      s2 += (double)part[(((size_t)oo * 2 + 1) * C + c) * inner + ii];
    }
  }
  block_sum_d(s1, s2, dummy, s_red);
  if (threadIdx.x != 0) return;
  const double mu = s1 / n;
  double var = s2 / n - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[c] * is;
  a[c] = av;
  b[c] = beta[c] - (float)mu * av;
  mean[c] = (float)mu;
  invstd[c] = is;
  if (running_mean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// ---- (2) BatchNorm state of z = W x from the Gram matrix of x: T = W G (Co x Ci), sx = sum x (Ci) --------------------
// one wave per output channel
__global__ __launch_bounds__(64) void gram_bn_finalize_kernel(
    const float* __restrict__ Tm, const float* __restrict__ W, const float* __restrict__ sx, int Co, int Ci,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, double n, float* __restrict__ a,
    float* __restrict__ b, float* __restrict__ mean, float* __restrict__ invstd, int centered) {
  // centered: Tm = W Gc with the centred Gram matrix Gc = sum (x - m)(x - m)^T, m = sx / n (eat_gram_centered): then
  // w^T Gc w = sum (z - mu)^2 = n var directly - no mu^2 subtracted from a sum of squares that is (mu/sigma)^2 larger
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < Ci; k += 64) {
    const double w = (double)W[(size_t)c * Ci + k];
    s1 += w * (double)sx[k];
    s2 += w * (double)Tm[(size_t)c * Ci + k];
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (threadIdx.x != 0) return;
  const double mu = s1 / n;
  double var = centered ? s2 / n : s2 / n - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[c] * is;
  a[c] = av;
  b[c] = beta[c] - (float)mu * av;
  mean[c] = (float)mu;
  invstd[c] = is;
  if (running_mean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// The same from the Gram matrix itself: row c of T = W G is formed here (fp64 accumulation; thread k walks column k of G
// with coalesced row reads - G need not be symmetric: any square matrix) and written out for the backward - the separate W G GEMM launch disappears.
// 256 threads = 64 columns x 4 slices of the j axis; dynamic LDS: 4 * Ci doubles.
__global__ __launch_bounds__(256) void gram_bn_finalize_g_kernel(
    const float* __restrict__ G, const float* __restrict__ W, const float* __restrict__ sx, int Co, int Ci,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, double n, float* __restrict__ Tm, float* __restrict__ a,
    float* __restrict__ b, float* __restrict__ mean, float* __restrict__ invstd, int centered) {
  extern __shared__ double s_part[];                      // [4][Ci]
  __shared__ double s_red[12];
  const int c = blockIdx.x, kx = threadIdx.x & 63, jp = threadIdx.x >> 6;
  const float* wr = W + (size_t)c * Ci;
  const int jn = (Ci + 3) / 4, j0 = jp * jn, j1 = (j0 + jn) < Ci ? (j0 + jn) : Ci;
  for (int k = kx; k < Ci; k += 64) {
    double t = 0.0;
    for (int j = j0; j < j1; ++j) t += (double)wr[j] * (double)G[(size_t)j * Ci + k];
    s_part[jp * Ci + k] = t;
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0, dummy = 0.0;
  for (int k = threadIdx.x; k < Ci; k += 256) {
    const double t = (s_part[k] + s_part[Ci + k]) + (s_part[2 * Ci + k] + s_part[3 * Ci + k]);
    Tm[(size_t)c * Ci + k] = (float)t;
    const double w = (double)wr[k];
    s1 += w * (double)sx[k];
    s2 += w * t;
  }
  block_sum_d(s1, s2, dummy, s_red);
  if (threadIdx.x != 0) return;
  const double mu = s1 / n;
  double var = centered ? s2 / n : s2 / n - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[c] * is;
  a[c] = av;
  b[c] = beta[c] - (float)mu * av;
  mean[c] = (float)mu;
  invstd[c] = is;
  if (running_mean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// ---- (3) backward coefficients of conv 1x1 -> BatchNorm(train) -> act, per output channel (one block each) -------------
//   in : W, Gx = sum g x^T, T = W G (Co x Ci each), sx (Ci), gpart [outer][Co][inner] (partials of sum g), a, mean, invstd
//   out: dW (Co x Ci), dgamma, dbeta (Co), e1 = a (m2 invstd mu - m1) (Co), and three Ci x Co transposes for the MFMA
//        linear kernel (which contracts over the contiguous axis): WaT = (diag(a) W)^T, WT = W^T,
//        W2T = -(diag(a m2 invstd) W)^T, so that M = W2T WT^T (Ci x Ci) and c0 = e1 WT^T (Ci)
//   frozen != 0: the layer normalised with fixed (running) statistics: m1 = m2 = 0 in dW / dx, dgamma / dbeta unchanged
__global__ __launch_bounds__(256) void expand_bwd_coef_kernel(
    const float* __restrict__ W, const float* __restrict__ Gx, const float* __restrict__ Tm,
    const float* __restrict__ sx, const float* __restrict__ gpart, int outer, int inner, int Co, int Ci,
    const float* __restrict__ a, const float* __restrict__ mean, const float* __restrict__ invstd, double n, int frozen,
    float* __restrict__ dW, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ WaT,
    float* __restrict__ WT, float* __restrict__ W2T, float* __restrict__ e1, int centered, float* __restrict__ e2_out) {
  // centered: Tm = W Gc with the centred Gram matrix Gc = sum (x - m)(x - m)^T (eat_gram_centered): T - mu sx^T IS W Gc -
  // the difference is taken term by term at accumulation time instead of between two (mu/sigma)^2-times larger sums.
  // (Gx - m1 sx^T and w.Gx - mu S1 lose only one factor mu/sigma, in fp64 here: Gx stays the plain sum g x^T.)
  __shared__ double s_red[12];
  const int c = blockIdx.x;
  double s1 = 0.0, wg = 0.0, dummy = 0.0;
  for (int e = threadIdx.x; e < outer * inner; e += blockDim.x) {
    const int o = e / inner, i = e - o * inner;
    s1 += (double)gpart[((size_t)o * Co + c) * inner + i];
  }
  for (int k = threadIdx.x; k < Ci; k += blockDim.x) wg += (double)W[(size_t)c * Ci + k] * (double)Gx[(size_t)c * Ci + k];
  block_sum_d(s1, wg, dummy, s_red);
  const double mu = (double)mean[c], is = (double)invstd[c], av = (double)a[c];
  const double s2 = is * (wg - mu * s1);                 // sum g xhat
  const double m1 = frozen ? 0.0 : s1 / n, m2 = frozen ? 0.0 : s2 / n;
  if (threadIdx.x == 0) {
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    e1[c] = (float)(av * (m2 * is * mu - m1));
  }
  const double e2 = av * m2 * is;
  if (threadIdx.x == 0 && e2_out) e2_out[c] = (float)e2;
  for (int k = threadIdx.x; k < Ci; k += blockDim.x) {
    const size_t idx = (size_t)c * Ci + k;
    const double sxk = (double)sx[k], w = (double)W[idx];
    const double tc = centered ? (double)Tm[idx] : (double)Tm[idx] - mu * sxk;
    dW[idx] = (float)(av * ((double)Gx[idx] - m1 * sxk - m2 * is * tc));
    if (WaT) {                                   // (NULL: the caller takes the packed form of eat_expand_bwd_wcat instead)
      const size_t tdx = (size_t)k * Co + c;
      WaT[tdx] = (float)(av * w);
      WT[tdx] = (float)w;
      W2T[tdx] = (float)(-e2 * w);
    }
  }
}

// ---- (3b) the operands of dx = [WaT | M] [g ; x] + c0 straight in MFMA-fragment order (round 6) -----------------------------
// Before: eat_expand_bwd_coef wrote three Ci x Co transposes, a prepack launch packed [W2T ; e1], a 1x1-conv launch formed
// [M ; c0] = [W2T ; e1] W, torch.cat glued [WaT | M] and another prepack launch packed it - five latency-bound launches per
// block per step (mn10: 14 blocks, ~0.67 ms + their boundaries).  Here ONE launch writes the pack of
//   Wcat (Ci x (Co + Ci)):  Wcat[i][k] = a[k] W[k][i]                       for k <  Co   (WaT)
//                           Wcat[i][Co + j] = -sum_c e2[c] W[c][i] W[c][j]  (M, exact fp32 MFMA over c)
// and c0[i] = sum_c e1[c] W[c][i].  Every pack element has exactly one owner (address = pure function of (i, k)), rows
// beyond Ci and the k padding are written as zeros.  kind: 0 fp32 fragments (eat_pw_prepack), 1 bf16, 2 bf16 hi + lo
// (eat_pw_prepack_bf16) - bit-identical layouts.
template <int KIND>
__device__ __forceinline__ void wcat_store(void* __restrict__ wp, int MT, int i, int k, float v) {
  const int mt = i >> 4;
  if constexpr (KIND == 0) {
    reinterpret_cast<float*>(wp)[((size_t)(k >> 2) * MT + mt) * 64 + (i & 15) + 16 * (k & 3)] = v;
  } else {
    constexpr int NP2 = KIND == 2 ? 2 : 1;
    const int r = k & 31;
    const size_t base = ((size_t)((k >> 5) * MT + mt) * NP2) * 512 + ((i & 15) + 16 * (r >> 3)) * 8 + (r & 7);
    __bf16* w16 = reinterpret_cast<__bf16*>(wp);
    const __bf16 hi = (__bf16)v;
    w16[base] = hi;
    if constexpr (KIND == 2) w16[base + 512] = (__bf16)(v - (float)hi);
  }
}

using wc_f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KIND>
__global__ __launch_bounds__(256) void expand_bwd_wcat_kernel(const float* __restrict__ W, const float* __restrict__ a,
                                                              const float* __restrict__ e2, const float* __restrict__ e1,
                                                              int Co, int Ci, int MT, int Kpad, void* __restrict__ wp,
                                                              float* __restrict__ c0) {
  __shared__ float s_acc[3][4][64];
  __shared__ float s_c0[16][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n_tiles = MT * MT;
  if ((int)blockIdx.x >= n_tiles) {
    // ---- elementwise part: WaT and the zero padding of the k axis
    const int I16 = MT * 16;
    const long long e = (long long)(blockIdx.x - n_tiles) * 256 + tid;
    const int kidx = (int)(e / I16), i = (int)(e - (long long)kidx * I16);
    const int n_k = Co + (Kpad - Co - Ci);
    if (kidx >= n_k) return;
    const int k = kidx < Co ? kidx : Ci + kidx;                       // (padding columns sit behind the M part)
    const float v = (kidx < Co && i < Ci) ? a[k] * W[(size_t)k * Ci + i] : 0.0f;
    wcat_store<KIND>(wp, MT, i, k, v);
    return;
  }
  // ---- one 16 x 16 tile of M: four waves split the reduction over the Co channels
  const int it = blockIdx.x / MT, jt = blockIdx.x - it * MT;
  const int i = it * 16 + (lane & 15), j = jt * 16 + (lane & 15), kq = lane >> 4;
  const bool iv = i < Ci, jv = j < Ci;
  const int steps = (Co + 3) / 4, per = (steps + 3) / 4;
  const int s0 = wv * per, s1 = (s0 + per) < steps ? (s0 + per) : steps;
  wc_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int sb = s0; sb < s1; sb += 8) {                                   // 8 k-steps of loads in flight per round
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = (sb + u) * 4 + kq;
      const bool ok = (sb + u) < s1 && c < Co;
      const int cc = ok ? c : 0;
      const float wi = W[(size_t)cc * Ci + (iv ? i : 0)], wj = W[(size_t)cc * Ci + (jv ? j : 0)];
      av[u] = (ok && iv) ? e2[cc] * wi : 0.0f;
      bv[u] = (ok && jv) ? wj : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
  if (wv > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_acc[wv - 1][r][lane] = acc[r];
  }
  // c0 of this row tile (first column tile only): 16 row groups x 16 columns
  float pc = 0.0f;
  if (jt == 0) {
    const int ci = it * 16 + (tid & 15), rg = tid >> 4;
    if (ci < Ci)
      for (int c = rg; c < Co; c += 16) pc = fmaf(e1[c], W[(size_t)c * Ci + ci], pc);
    s_c0[rg][tid & 15] = pc;
  }
  __syncthreads();
  if (jt == 0 && tid < 16) {
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += s_c0[r][tid];
    if (it * 16 + tid < Ci) c0[it * 16 + tid] = t;
  }
  if (wv != 0) return;
  // C / D layout of the MFMA: lane (kq, n) holds rows 4 kq + r of column n
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = acc[r] + s_acc[0][r][lane] + s_acc[1][r][lane] + s_acc[2][r][lane];
    const int mi = it * 16 + kq * 4 + r, mj = jt * 16 + (lane & 15);
    if (mj < Ci) wcat_store<KIND>(wp, MT, mi, Co + mj, (mi < Ci) ? -v : 0.0f);
  }
}

}  // namespace

namespace eat {
int bn_stats_partial(const float* z, int B, int C, int S, float* part, hipStream_t s) {
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((unsigned)(B * C)), dim3(S >= 1024 ? 256 : 64), 0, s, z, C, S, part);
  return check_launch("bn_stats_partial");
}
int act_grad_sum(const float* dy, const float* z, const float* a, const float* b, int act, float* g, float* gpart, int B,
                 int C, int S, hipStream_t s) {
  hipLaunchKernelGGL(act_grad_sum_kernel, dim3((unsigned)(B * C)), dim3(S >= 1024 ? 256 : 64), 0, s, dy, z, a, b, act, g,
                     gpart, C, S);
  return check_launch("act_grad_sum");
}
}  // namespace eat

extern "C" int eat_bn_stats_partial(const float* z, int B, int C, int S, float* part, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_bn_stats_partial: bad shape");
  return eat::bn_stats_partial(z, B, C, S, part, (hipStream_t)stream);
}

extern "C" int eat_act_grad_sum(const float* dy, const float* z, const float* a, const float* b, int act, float* g,
                                float* gpart, int B, int C, int S, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_act_grad_sum: bad act %d", act);
  if (B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_act_grad_sum: bad shape");
  return eat::act_grad_sum(dy, z, a, b, act, g, gpart, B, C, S, (hipStream_t)stream);
}

template <typename ZT>
static int se_bn_bwd_partials_impl(const ZT* d, const ZT* z, const float* a, const float* b, const float* mean, float* P, int B,
                                   int C, int S, int act, hipStream_t stream, const char* what) {
  if (S <= 512 && (S & 3) == 0) {
    // small planes: waves own planes (two per wave for S <= 128); ~8 planes per wave fill the chip several times over
    const int n_planes = B * C, ppb = S <= 128 ? 8 : 4;            // planes per block per grid-stride step
    int nb = (n_planes + ppb * 8 - 1) / (ppb * 8);
    nb = nb < 1 ? 1 : nb;
    if (S <= 128) {
      EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((se_bn_bwd_partials_small_kernel<ACT, 32, ZT>), dim3((unsigned)nb), dim3(256), 0,
                                               stream, d, z, a, b, mean, P, C, S, n_planes));
    } else {
      EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((se_bn_bwd_partials_small_kernel<ACT, 64, ZT>), dim3((unsigned)nb), dim3(256), 0,
                                               stream, d, z, a, b, mean, P, C, S, n_planes));
    }
    return eat::check_launch(what);
  }
  const dim3 blk(S >= 1024 ? 256 : 64);
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((se_bn_bwd_partials_kernel<ACT, ZT>), dim3((unsigned)(B * C)), blk, 0, stream, d, z, a,
                                           b, mean, P, C, S, B * C));
  return eat::check_launch(what);
}

extern "C" int eat_se_bn_bwd_partials(const float* d, const float* z, const float* a, const float* b, const float* mean,
                                      float* P, int B, int C, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_se_bn_bwd_partials: bad act %d", act);
  if (B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_se_bn_bwd_partials: bad shape");
  return se_bn_bwd_partials_impl<float>(d, z, a, b, mean, P, B, C, S, act, (hipStream_t)stream, "eat_se_bn_bwd_partials");
}

// the same over bf16-stored d and z (act_io.h: the bf16-storage plan of BASELINE configs[2])
extern "C" int eat_se_bn_bwd_partials_b16(const void* d, const void* z, const float* a, const float* b, const float* mean,
                                          float* P, int B, int C, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_se_bn_bwd_partials_b16: bad act %d", act);
  if (!d || !z || B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_se_bn_bwd_partials_b16: bad shape");
  return se_bn_bwd_partials_impl<eat::bf16_t>(reinterpret_cast<const eat::bf16_t*>(d), reinterpret_cast<const eat::bf16_t*>(z), a,
                                              b, mean, P, B, C, S, act, (hipStream_t)stream, "eat_se_bn_bwd_partials_b16");
}

extern "C" int eat_se_bn_bwd_combine(const float* P, const float* gscale, const float* gadd, const float* invstd, int B,
                                     int C, double* sums, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1) return eat::fail(EAT_EINVAL, "eat_se_bn_bwd_combine: bad shape");
  hipLaunchKernelGGL(se_bn_bwd_combine_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, P, gscale, gadd, invstd,
                     B, C, sums);
  return eat::check_launch("eat_se_bn_bwd_combine");
}

// ---- channel sums of a BatchNorm backward from the tile partials of pw_epilogue_gstats ([tiles][2][C]: sum g, sum g z) ----
// sums[c] = sum g, sums[C + c] = invstd[c] (sum g z - mean[c] sum g)   (fp64; the layout eat_bn_act_bwd_apply /
// eat_dw_conv_bwd_bn_g read).  rows: the tile partials themselves (PT = float) or their row-group sums (PT = double).
template <typename PT>
__global__ __launch_bounds__(256) void bn_bwd_sums_finish_kernel(const PT* __restrict__ v, int rows, int C,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, double* __restrict__ sums,
                                                                 const float* __restrict__ ga, const float* __restrict__ gb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  int r = 0;
  for (; r + 3 < rows; r += 4) {                       // (loads of four rows in flight: rows <= 64 here)
    const PT a0 = v[(size_t)r * 2 * C + c], a1 = v[(size_t)(r + 1) * 2 * C + c];
    const PT a2 = v[(size_t)(r + 2) * 2 * C + c], a3 = v[(size_t)(r + 3) * 2 * C + c];
    const PT b0 = v[(size_t)r * 2 * C + C + c], b1 = v[(size_t)(r + 1) * 2 * C + C + c];
    const PT b2 = v[(size_t)(r + 2) * 2 * C + C + c], b3 = v[(size_t)(r + 3) * 2 * C + C + c];
    s0 += (double)a0; s0 += (double)a1; s0 += (double)a2; s0 += (double)a3;
    s1 += (double)b0; s1 += (double)b1; s1 += (double)b2; s1 += (double)b3;
  }
  for (; r < rows; ++r) {
    s0 += (double)v[(size_t)r * 2 * C + c];
    s1 += (double)v[(size_t)r * 2 * C + C + c];
  }
  sums[c] = s0;
  // ga / gb (the BatchNorm's a, b): the partials hold sum g (z - c), c = -b / a (pw_epilogue.h: gstat_center) - otherwise sum g z
  const double ctr = ga ? (double)eat::gstat_center(ga[c], gb[c]) : 0.0;
  sums[C + c] = (double)invstd[c] * (s1 + (ctr - (double)mean[c]) * s0);
}

// doubles of workspace eat_bn_finalize_partials wants for this shape (0: none - few partial rows)
extern "C" int eat_bn_finalize_ws_doubles(int outer, int C, int inner) {
  if (outer < 1 || C < 1 || inner < 1) return 0;
  return fin_groups(outer, C, inner) * 2 * C * inner;
}

extern "C" int eat_bn_finalize_partials(const float* part, int outer, int C, int inner, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float momentum,
                                        float eps, double n, float* a, float* b, float* mean, float* invstd, double* ws,
                                        eat_stream_t stream) {
  eat::clear_stale_error();
  if (outer < 1 || C < 1 || inner < 1) return eat::fail(EAT_EINVAL, "eat_bn_finalize_partials: bad shape");
  const int G = ws ? fin_groups(outer, C, inner) : 0;
  if (G > 0) {
    const int W = 2 * C * inner, cw = fin_col_threads(W);
    hipLaunchKernelGGL(bn_partials_rowgroups_kernel, dim3((unsigned)((W + cw - 1) / cw), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, part, outer, W, ws);
    hipLaunchKernelGGL(bn_finalize_partials_kernel<double>, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream,
                       (const double*)ws, G, C, inner, gamma, beta, running_mean, running_var, momentum, eps, n, a, b, mean,
                       invstd);
  } else {
    hipLaunchKernelGGL(bn_finalize_partials_kernel<float>, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, part, outer,
                       C, inner, gamma, beta, running_mean, running_var, momentum, eps, n, a, b, mean, invstd);
  }
  return eat::check_launch("eat_bn_finalize_partials");
}

// row groups of the tile partials: from 32 tiles on (one thread walking 504 rows of its channel took 200 us)
static int bwd_tile_groups(int tiles) { return tiles < 32 ? 0 : (tiles / 8 > kFinGroupsMax ? kFinGroupsMax : tiles / 8); }
extern "C" int eat_bn_bwd_sums_ws_doubles(int tiles, int C) { return tiles < 1 || C < 1 ? 0 : bwd_tile_groups(tiles) * 2 * C; }

// part: [tiles][2][C] from eat_pw_conv_gstats_fwd; ws: eat_bn_bwd_sums_ws_doubles(tiles, C) doubles (or NULL when that is 0)
extern "C" int eat_bn_bwd_sums_from_tiles(const float* part, int tiles, int C, const float* mean, const float* invstd,
                                          const float* g_a, const float* g_b, double* ws, double* sums, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!part || !mean || !invstd || !sums || tiles < 1 || C < 1 || ((g_a == nullptr) != (g_b == nullptr)))
    return eat::fail(EAT_EINVAL, "eat_bn_bwd_sums_from_tiles: bad arguments");
  const int G = bwd_tile_groups(tiles);
  if (G > 0 && !ws) return eat::fail(EAT_EINVAL, "eat_bn_bwd_sums_from_tiles: %d tiles need the row-group workspace", tiles);
  const dim3 grid((unsigned)((C + 255) / 256));
  if (G > 0) {
    const int W = 2 * C, cw = fin_col_threads(W);
    hipLaunchKernelGGL(bn_partials_rowgroups_kernel, dim3((unsigned)((W + cw - 1) / cw), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, part, tiles, W, ws);
    hipLaunchKernelGGL(bn_bwd_sums_finish_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)ws, G, C, mean,
                       invstd, sums, g_a, g_b);
  } else {
    hipLaunchKernelGGL(bn_bwd_sums_finish_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, part, tiles, C, mean, invstd,
                       sums, g_a, g_b);
  }
  return eat::check_launch("eat_bn_bwd_sums_from_tiles");
}

extern "C" int eat_gram_bn_finalize(const float* Tm, const float* W, const float* sx, int Co, int Ci, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                    double n, float* a, float* b, float* mean, float* invstd, int centered,
                                    eat_stream_t stream) {
  eat::clear_stale_error();
  if (Co < 1 || Ci < 1) return eat::fail(EAT_EINVAL, "eat_gram_bn_finalize: bad shape");
  hipLaunchKernelGGL(gram_bn_finalize_kernel, dim3((unsigned)Co), dim3(64), 0, (hipStream_t)stream, Tm, W, sx, Co, Ci, gamma,
                     beta, running_mean, running_var, momentum, eps, n, a, b, mean, invstd, centered);
  return eat::check_launch("eat_gram_bn_finalize");
}

extern "C" int eat_gram_bn_finalize_g(const float* G, const float* W, const float* sx, int Co, int Ci, const float* gamma,
                                      const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                      double n, float* Tm, float* a, float* b, float* mean, float* invstd, int centered,
                                      eat_stream_t stream) {
  eat::clear_stale_error();
  if (Co < 1 || Ci < 1 || Ci > 2048) return eat::fail(EAT_EINVAL, "eat_gram_bn_finalize_g: bad shape (%d x %d)", Co, Ci);
  hipLaunchKernelGGL(gram_bn_finalize_g_kernel, dim3((unsigned)Co), dim3(256), (size_t)4 * Ci * sizeof(double),
                     (hipStream_t)stream, G, W, sx, Co, Ci, gamma, beta, running_mean, running_var, momentum, eps, n, Tm, a, b,
                     mean, invstd, centered);
  return eat::check_launch("eat_gram_bn_finalize_g");
}

extern "C" int eat_expand_bwd_coef(const float* W, const float* Gx, const float* Tm, const float* sx, const float* gpart,
                                   int outer, int inner, int Co, int Ci, const float* a, const float* mean,
                                   const float* invstd, double n, int frozen, float* dW, float* dgamma, float* dbeta,
                                   float* WaT, float* WT, float* W2T, float* e1, int centered, float* e2, eat_stream_t stream) {
  eat::clear_stale_error();
  if (Co < 1 || Ci < 1 || outer < 1 || inner < 1) return eat::fail(EAT_EINVAL, "eat_expand_bwd_coef: bad shape");
  if ((WaT == nullptr) != (WT == nullptr) || (WaT == nullptr) != (W2T == nullptr))
    return eat::fail(EAT_EINVAL, "eat_expand_bwd_coef: the three transposes come together (or none)");
  hipLaunchKernelGGL(expand_bwd_coef_kernel, dim3((unsigned)Co), dim3(256), 0, (hipStream_t)stream, W, Gx, Tm, sx, gpart,
                     outer, inner, Co, Ci, a, mean, invstd, n, frozen, dW, dgamma, dbeta, WaT, WT, W2T, e1, centered, e2);
  return eat::check_launch("eat_expand_bwd_coef");
}

// floats (kind 0) / bf16 elements (kinds 1, 2) of the pack eat_expand_bwd_wcat writes
extern "C" int eat_expand_bwd_wcat_elems(int Co, int Ci, int kind) {
  if (Co < 1 || Ci < 1 || kind < 0 || kind > 2) return 0;
  const long long MT = (Ci + 15) / 16, K = (long long)Co + Ci;
  const long long n = kind == 0 ? (K / 4) * MT * 64 : ((K + 31) / 32) * MT * (kind == 2 ? 2 : 1) * 512;
  return n > 0x7fffffffLL ? 0 : (int)n;
}

extern "C" int eat_expand_bwd_wcat(const float* W, const float* a, const float* e2, const float* e1, int Co, int Ci, int kind,
                                   void* wp, float* c0, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!W || !a || !e2 || !e1 || !wp || !c0 || Co < 1 || Ci < 1 || kind < 0 || kind > 2)
    return eat::fail(EAT_EINVAL, "eat_expand_bwd_wcat: bad arguments");
  if (((Co + Ci) & 3) != 0) return eat::fail(EAT_EINVAL, "eat_expand_bwd_wcat: Co + Ci = %d must be a multiple of 4", Co + Ci);
  const int MT = (Ci + 15) / 16;
  const int Kpad = kind == 0 ? Co + Ci : ((Co + Ci + 31) / 32) * 32;
  const long long ew = (long long)MT * 16 * (Co + (Kpad - Co - Ci));
  const unsigned grid = (unsigned)(MT * MT + (ew + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(expand_bwd_wcat_kernel<0>, dim3(grid), dim3(256), 0, s, W, a, e2, e1, Co, Ci, MT, Kpad, wp, c0);
  else if (kind == 1) hipLaunchKernelGGL(expand_bwd_wcat_kernel<1>, dim3(grid), dim3(256), 0, s, W, a, e2, e1, Co, Ci, MT, Kpad, wp, c0);
  else hipLaunchKernelGGL(expand_bwd_wcat_kernel<2>, dim3(grid), dim3(256), 0, s, W, a, e2, e1, Co, Ci, MT, Kpad, wp, c0);
  return eat::check_launch("eat_expand_bwd_wcat");
}
