// Training-step kernels for gfx950: batch-statistics BatchNorm (forward and backward, with the
// activation folded in), squeeze-excitation reductions, and the weight/data gradients of the
// depthwise and pointwise convolutions.
//   reference semantics: nn.BatchNorm2d(eps=1e-3, momentum=0.01) in train mode
//   (models/mn/model.py:114-115), nn.Hardswish / nn.ReLU, autograd of F.conv2d
//   (SURVEY.md Appendix C lists the formulas the reference leaves to autograd).
// Round-1 structure: every pass is its own streaming kernel over (B, C, S) planes (one workgroup
// per plane, float4 along the time axis, wave-shuffle + LDS block reduction, per-channel totals
// accumulated in fp64 atomics so that sums over up to 8M elements do not lose precision).
#include <cstdlib>
#include "eat_common.h"
#include "act_io.h"

namespace {

using eat::Io;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Optional transform of the x operand of the 1x1 weight gradient: x' = act(a[ci] * x + b[ci]) evaluated on load (training:
// the project conv read BN + act of the depthwise output on load, so the activated tensor does not exist; mn_train.py)
// actr (Co) or NULL: additive constant of the dz operand, per row, applied to loaded elements only - with a = 1, b = actr =
// -mean and dz == x the kernels form the CENTRED Gram matrix sum (x - m)(x - m)^T (eat_gram_centered)
struct WgTf { const float* a; const float* b; int act; const float* actr = nullptr; };
// atomic add on a pointer KNOWN to be global memory (hipcc cannot always infer the address space of a pointer offset by a
// run-time slot index, and its expansion of a flat fp32 atomic fails on gfx950: "Operand has incorrect register class")
__device__ __forceinline__ void global_atomic_add(float* p, float v) {
  typedef __attribute__((address_space(1))) float gfloat;
  __hip_atomic_fetch_add((gfloat*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// branch-free: act(u) = max(u, lo) * clamp(u * ca + cb, 0, 1) with (lo, ca, cb) = none (-inf, 0, 1), ReLU (0, 0, 1),
// Hardswish (-inf, 1/6, 1/2) - the weight-gradient kernels are VALU-bound on the bf16 hi/lo split already
__device__ __forceinline__ float wg_tf(float v, float a, float b, int act) {
  const float u = fmaf(a, v, b);
  if (act == EAT_ACT_NONE) return u;                       // (wave-uniform) the centring transform of the Gram / Gx launches
  const float lo = act == EAT_ACT_RELU ? 0.0f : -__builtin_huge_valf();
  const float ca = act == EAT_ACT_HSWISH ? (1.0f / 6.0f) : 0.0f, cb = act == EAT_ACT_HSWISH ? 0.5f : 1.0f;
  return fmaxf(u, lo) * __builtin_amdgcn_fmed3f(fmaf(u, ca, cb), 0.0f, 1.0f);
}

template <int ACT>
__device__ __forceinline__ float act_grad(float u) {   // d act(u) / du  (PyTorch conventions)
  if constexpr (ACT == EAT_ACT_RELU) return u > 0.0f ? 1.0f : 0.0f;
  if constexpr (ACT == EAT_ACT_HSWISH) return u < -3.0f ? 0.0f : (u <= 3.0f ? u * (1.0f / 3.0f) + 0.5f : 1.0f);
  return 1.0f;
}

// block-wide sum of two values; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b, float* s_red) {
  a = eat::wave_sum(a);
  b = eat::wave_sum(b);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { s_red[wv] = a; s_red[8 + wv] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < nw; ++i) { ta += s_red[i]; tb += s_red[8 + i]; }
    a = ta; b = tb;
  }
}

// ---- per-channel sum / sum of squares of z (B,C,S) -----------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int C, int S,
                                                       double* __restrict__ sums) {
  __shared__ float s_red[16];
  const int plane = blockIdx.x, c = plane % C;
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
  block_sum2(s1, s2, s_red);
  if (threadIdx.x == 0) {
    atomicAdd(sums + c, (double)s1);
    atomicAdd(sums + C + c, (double)s2);
  }
}

// Small planes (late stages: 8x63, 4x32 positions): with one block per plane the 2 x B x C fp64 atomics on C addresses
// are the whole cost (~50 us per launch for 20 MB tensors, 46 launches per step).  Here a block owns PPB samples of ONE
// channel: the same coalesced float4 reads, PPB times fewer atomics per channel.
__global__ __launch_bounds__(256) void bn_stats_multi_kernel(const float* __restrict__ z, int B, int C, int S4, int PPB,
                                                             double* __restrict__ sums) {
  __shared__ float s_red[16];
  const int c = blockIdx.x, b0 = blockIdx.y * PPB;
  const int nb = (B - b0) < PPB ? (B - b0) : PPB;
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < nb * S4; e += 256) {
    const int bl = e / S4, i = e - bl * S4;
    const float4 v = *reinterpret_cast<const float4*>(z + ((size_t)(b0 + bl) * C + c) * (4 * S4) + 4 * i);
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  block_sum2(s1, s2, s_red);
  if (threadIdx.x == 0) {
    atomicAdd(sums + c, (double)s1);
    atomicAdd(sums + C + c, (double)s2);
  }
}

// ---- finalize: batch mean / biased var -> affine (a, b), saved (mean, invstd), running buffers --
__global__ void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps, double n, int C,
                                   float* __restrict__ a, float* __restrict__ b, float* __restrict__ mean,
                                   float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = sums[c] / n;
  double var = sums[C + c] / n - mu * mu;
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

// ---- y = act(a_c z + b_c) [+ res]; optional per-(b,c) sums of y (SE squeeze / head pool) ----------
// ZT / YT: storage types of z / y (act_io.h; bf16 in the bf16-storage plan: the pool sums the values as STORED; ZT = bf16 with
// YT = float: the project conv's BatchNorm - z_p is stored in bf16, the block output it produces is an fp32 tensor)
template <int ACT, typename ZT = float, typename YT = ZT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const ZT* __restrict__ z, const float* __restrict__ a,
                                                         const float* __restrict__ b, const float* __restrict__ res,
                                                         YT* __restrict__ y, float* __restrict__ pool, int C, int S,
                                                         eat::bf16_t* __restrict__ y16 = nullptr) {
  // y16: optional bf16 COPY of an fp32 y (the block output of the bf16-storage plan: the next block's expand conv reads the
  // copy - bit-identical to reading y, the conv rounds its operand the same way - at half the operand traffic)
  __shared__ float s_red[16];
  const int plane = blockIdx.x, c = plane % C;
  const float av = a[c], bv = b[c];
  const size_t base = (size_t)plane * S;
  float ps = 0.f, dummy = 0.f;
  if ((S & 3) == 0) {
#pragma unroll 4
    for (int i = threadIdx.x * 4; i < S; i += blockDim.x * 4) {
      const float4 v = Io<ZT>::load4(z + base + i);
      float4 o = make_float4(eat::activate<ACT>(fmaf(av, v.x, bv)), eat::activate<ACT>(fmaf(av, v.y, bv)),
                             eat::activate<ACT>(fmaf(av, v.z, bv)), eat::activate<ACT>(fmaf(av, v.w, bv)));
      if (res) {
        const float4 r = *reinterpret_cast<const float4*>(res + base + i);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if constexpr (Io<YT>::kBf) { o.x = eat::bf_round(o.x); o.y = eat::bf_round(o.y); o.z = eat::bf_round(o.z); o.w = eat::bf_round(o.w); }
      if (y) Io<YT>::store4(y + base + i, o);
      if (y16) Io<eat::bf16_t>::store4(y16 + base + i, o);
      ps += (o.x + o.y) + (o.z + o.w);
    }
  } else {
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
      float o = Io<YT>::rnd(eat::activate<ACT>(fmaf(av, Io<ZT>::load1(z + base + i), bv)) + (res ? res[base + i] : 0.0f));
      if (y) Io<YT>::store1(y + base + i, o);
      if (y16) Io<eat::bf16_t>::store1(y16 + base + i, o);
      ps += o;
    }
  }
  if (pool) {
    block_sum2(ps, dummy, s_red);
    if (threadIdx.x == 0) pool[plane] = ps;       // one block per plane: plain store, no atomics
  }
}

// g = (dy * gscale[b,c] + gadd[b,c]) * act'(a z + b);  xhat = (z - mean) * invstd
template <int ACT>
__device__ __forceinline__ float grad_pre(float dy, float zv, float av, float bv, float gs, float ga) {
  return fmaf(dy, gs, ga) * act_grad<ACT>(fmaf(av, zv, bv));
}

// ---- backward pass 1: per-channel sum g and sum g*xhat -------------------------------------------------
template <int ACT, typename ZT = float, typename DT = ZT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(
    const DT* __restrict__ dy, const ZT* __restrict__ z, const float* __restrict__ a,
    const float* __restrict__ b, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gscale, const float* __restrict__ gadd, int C, int S, double* __restrict__ sums) {
  __shared__ float s_red[16];
  const int plane = blockIdx.x, c = plane % C;
  const float av = a[c], bv = b[c], mu = mean[c], is = invstd[c];
  const float gs = gscale ? gscale[plane] : 1.0f, ga = gadd ? gadd[plane] : 0.0f;
  const size_t base = (size_t)plane * S;
  float s1 = 0.f, s2 = 0.f;
  if ((S & 3) == 0) {
#pragma unroll 4
    for (int i = threadIdx.x * 4; i < S; i += blockDim.x * 4) {
      const float4 d = Io<DT>::load4(dy + base + i);
      const float4 v = Io<ZT>::load4(z + base + i);
      const float g0 = grad_pre<ACT>(d.x, v.x, av, bv, gs, ga), g1 = grad_pre<ACT>(d.y, v.y, av, bv, gs, ga);
      const float g2 = grad_pre<ACT>(d.z, v.z, av, bv, gs, ga), g3 = grad_pre<ACT>(d.w, v.w, av, bv, gs, ga);
      s1 += (g0 + g1) + (g2 + g3);
      s2 += (g0 * (v.x - mu) + g1 * (v.y - mu)) + (g2 * (v.z - mu) + g3 * (v.w - mu));
    }
  } else {
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
      const float zv = Io<ZT>::load1(z + base + i);
      const float g = grad_pre<ACT>(Io<DT>::load1(dy + base + i), zv, av, bv, gs, ga);
      s1 += g;
      s2 += g * (zv - mu);
    }
  }
  s2 *= is;
  block_sum2(s1, s2, s_red);
  if (threadIdx.x == 0) {
    atomicAdd(sums + c, (double)s1);
    atomicAdd(sums + C + c, (double)s2);
  }
}

// small planes: one block per (channel, PPB samples), see bn_stats_multi_kernel
template <int ACT, typename ZT = float, typename DT = ZT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_multi_kernel(
    const DT* __restrict__ dy, const ZT* __restrict__ z, const float* __restrict__ a,
    const float* __restrict__ b, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gscale, const float* __restrict__ gadd, int B, int C, int S4, int PPB,
    double* __restrict__ sums) {
  __shared__ float s_red[16];
  const int c = blockIdx.x, b0 = blockIdx.y * PPB;
  const int nb = (B - b0) < PPB ? (B - b0) : PPB;
  const float av = a[c], bv = b[c], mu = mean[c], is = invstd[c];
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < nb * S4; e += 256) {
    const int bl = e / S4, i = e - bl * S4;
    const size_t plane = (size_t)(b0 + bl) * C + c;
    const float gs = gscale ? gscale[plane] : 1.0f, ga = gadd ? gadd[plane] : 0.0f;
    const float4 d = Io<DT>::load4(dy + plane * (4 * S4) + 4 * i);
    const float4 v = Io<ZT>::load4(z + plane * (4 * S4) + 4 * i);
    const float g0 = grad_pre<ACT>(d.x, v.x, av, bv, gs, ga), g1 = grad_pre<ACT>(d.y, v.y, av, bv, gs, ga);
    const float g2 = grad_pre<ACT>(d.z, v.z, av, bv, gs, ga), g3 = grad_pre<ACT>(d.w, v.w, av, bv, gs, ga);
    s1 += (g0 + g1) + (g2 + g3);
    s2 += (g0 * (v.x - mu) + g1 * (v.y - mu)) + (g2 * (v.z - mu) + g3 * (v.w - mu));
  }
  s2 *= is;
  block_sum2(s1, s2, s_red);
  if (threadIdx.x == 0) {
    atomicAdd(sums + c, (double)s1);
    atomicAdd(sums + C + c, (double)s2);
  }
}

// ---- backward pass 2: dz = a * (g - sum_g/N - xhat * sum_gx/N) ----------------------------------------------
// DT: storage type of dy AND dz (fp32; bf16: the expand BatchNorm of a DyMN block under the bf16-storage plan, g_e -> dz_e in place)
template <int ACT, typename ZT = float, typename DT = float>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const DT* __restrict__ dy, const ZT* __restrict__ z, const float* __restrict__ a,
    const float* __restrict__ b, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gscale, const float* __restrict__ gadd, const double* __restrict__ sums,
    DT* __restrict__ dz, int C, int S, double n, eat::bf16_t* __restrict__ dz16 = nullptr) {
  // dz16: optional bf16 COPY of dz (what the data-gradient 1x1 conv of the bf16-storage plan reads: see bn_act_fwd_kernel)
  const int plane = blockIdx.x, c = plane % C;
  const float av = a[c], bv = b[c], mu = mean[c], is = invstd[c];
  const float m1 = (float)(sums[c] / n), m2 = (float)(sums[C + c] / n);
  const float gs = gscale ? gscale[plane] : 1.0f, ga = gadd ? gadd[plane] : 0.0f;
  const size_t base = (size_t)plane * S;
  auto f = [&](float d, float v) {
    const float g = grad_pre<ACT>(d, v, av, bv, gs, ga);
    return av * (g - m1 - (v - mu) * is * m2);
  };
  if ((S & 3) == 0) {
#pragma unroll 4
    for (int i = threadIdx.x * 4; i < S; i += blockDim.x * 4) {
      const float4 d = Io<DT>::load4(dy + base + i);
      const float4 v = Io<ZT>::load4(z + base + i);
      const float4 o = make_float4(f(d.x, v.x), f(d.y, v.y), f(d.z, v.z), f(d.w, v.w));
      Io<DT>::store4(dz + base + i, o);
      if (dz16) Io<eat::bf16_t>::store4(dz16 + base + i, o);
    }
  } else {
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
      const float o = f(Io<DT>::load1(dy + base + i), Io<ZT>::load1(z + base + i));
      Io<DT>::store1(dz + base + i, o);
      if (dz16) Io<eat::bf16_t>::store1(dz16 + base + i, o);
    }
  }
}

// ---- out[b,c] = sum_s u[b,c,s] * v'[b,c,s], v' = v or act(a_c v + b_c) (SE: d scale) ---------------------------------
template <int ACT>
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ out, int C, int S) {
  __shared__ float s_red[16];
  const int plane = blockIdx.x, c = plane % C;
  const float av = a ? a[c] : 1.0f, bv = b ? b[c] : 0.0f;
  const size_t base = (size_t)plane * S;
  float s1 = 0.f, dummy = 0.f;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    float t = v[base + i];
    if (a) t = eat::activate<ACT>(fmaf(av, t, bv));
    s1 += u[base + i] * t;
  }
  block_sum2(s1, dummy, s_red);
  if (threadIdx.x == 0) out[plane] = s1;
}

// ---- depthwise data gradient: dx[c,i,j] = sum_{u,v} w[c,u,v] dz[c,(i+p-u)/s,(j+p-v)/s] (+ res) ----------
template <int K, int STRIDE>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                       const float* __restrict__ res, float* __restrict__ dx,
                                                       int C, int F, int T, int Fo, int To, int per_plane_w) {
  constexpr int P = (K - 1) / 2;
  const int plane = blockIdx.y, c = plane % C;
  float wr[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) wr[i] = w[(size_t)(per_plane_w ? plane : c) * K * K + i];
  const float* g = dz + (size_t)plane * Fo * To;
  const size_t base = (size_t)plane * F * T;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * T; e += gridDim.x * blockDim.x) {
    const int i = e / T, j = e - i * T;
    float acc = res ? res[base + e] : 0.0f;
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const int ii = i + P - u;
      if (ii < 0 || (ii % STRIDE) != 0) continue;
      const int io = ii / STRIDE;
      if (io >= Fo) continue;
#pragma unroll
      for (int v = 0; v < K; ++v) {
        const int jj = j + P - v;
        if (jj < 0 || (jj % STRIDE) != 0) continue;
        const int jo = jj / STRIDE;
        if (jo < To) acc = fmaf(wr[u * K + v], g[(size_t)io * To + jo], acc);
      }
    }
    dx[base + e] = acc;
  }
}

// ---- depthwise data gradient, stride 2, sliding form: one thread per dx column j walking down the
// rows; only the taps whose parity matches contribute (<= ceil(K/2)^2 loads per element instead of
// K*K predicated iterations), lanes on consecutive j read dz at half stride (coalesced).
template <int K>
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                          const float* __restrict__ res, float* __restrict__ dx,
                                                          int n_planes, int C, int F, int T, int Fo, int To,
                                                          int per_plane_w) {
  // Polyphase form: dx[i][j] = sum over the taps (u, v) with (i + P - u), (j + P - v) even of w[u][v] dz[(i+P-u)/2][(j+P-v)/2].
  // The column parity of a thread is fixed, so its tap columns v = v0 + 2q are selected ONCE into registers (the round-1
  // kernel indexed the tap array with run-time (u, v) inside the row loop); the rows are walked in pairs (2m, 2m+1), whose
  // tap rows are compile-time constants, over a sliding window of dz rows m-1, m, m+1 - every dz row is loaded once per
  // thread, one row ahead of its use.  Writes are coalesced 256-byte row segments per wave.
  constexpr int P = (K - 1) / 2;
  constexpr int NV = (K + 1) / 2;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int plane = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (plane >= n_planes || j >= T) return;
  const int c = plane % C;
  const float* wp = w + (size_t)(per_plane_w ? plane : c) * K * K;
  const int v0 = (j + P) & 1;
  float ws[K][NV];
  int jo[NV];
  bool jok[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int v = v0 + 2 * q;
    jo[q] = (j + P - v) >> 1;
    jok[q] = v < K && (j + P - v) >= 0 && jo[q] < To;
    if (!jok[q]) jo[q] = 0;
#pragma unroll
    for (int u = 0; u < K; ++u) ws[u][q] = v < K ? wp[u * K + v] : 0.0f;
  }
  const float* g = dz + (size_t)plane * Fo * To;
  auto load_row = [&](int io, float (&r)[NV]) {
    const bool rok = io >= 0 && io < Fo;
    const float* row = g + (size_t)(rok ? io : 0) * To;
#pragma unroll
    for (int q = 0; q < NV; ++q) r[q] = (rok && jok[q]) ? row[jo[q]] : 0.0f;
  };
  const size_t base = (size_t)plane * F * T + j;
  float rm[NV], r0[NV], r1[NV], r2[NV];        // dz rows m-1, m, m+1 and the prefetched m+2
#pragma unroll
  for (int q = 0; q < NV; ++q) rm[q] = 0.0f;
  load_row(0, r0);
  load_row(1, r1);
  for (int m = 0; 2 * m < F; ++m) {
    load_row(m + 2, r2);
    const int ie = 2 * m, io_ = 2 * m + 1;
    float ae = res ? res[base + (size_t)ie * T] : 0.0f;
    float ao = (res && io_ < F) ? res[base + (size_t)io_ * T] : 0.0f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      if constexpr (K == 3) {
        ae = fmaf(ws[1][q], r0[q], ae);                                 // row 2m:   u = 1 -> dz row m
        ao = fmaf(ws[0][q], r1[q], fmaf(ws[2][q], r0[q], ao));          // row 2m+1: u = 0 -> m+1, u = 2 -> m
      } else {
        ae = fmaf(ws[0][q], r1[q], fmaf(ws[2][q], r0[q], fmaf(ws[4][q], rm[q], ae)));   // u = 0, 2, 4 -> m+1, m, m-1
        ao = fmaf(ws[1][q], r1[q], fmaf(ws[3][q], r0[q], ao));                          // u = 1, 3    -> m+1, m
      }
    }
    dx[base + (size_t)ie * T] = ae;
    if (io_ < F) dx[base + (size_t)io_ * T] = ao;
#pragma unroll
    for (int q = 0; q < NV; ++q) { rm[q] = r0[q]; r0[q] = r1[q]; r1[q] = r2[q]; }
  }
}

// ---- depthwise / stem weight gradient: dw[c,u,v] = sum_{b,i,j} dz[b,c,i,j] x[b,cx,i*s+u-p,j*s+v-p] -------
// One block per (channel, batch slice); x has XC channels (XC == C depthwise, XC == 1 stem).
template <int K, int STRIDE>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                       float* __restrict__ dw, int B, int C, int XC, int F, int T,
                                                       int Fo, int To, int b_per_block, int per_sample) {
  constexpr int P = (K - 1) / 2;
  __shared__ float s_red[4][K * K];
  const int c = blockIdx.x, b0 = blockIdx.y * b_per_block;
  const int b1 = (b0 + b_per_block) < B ? (b0 + b_per_block) : B;
  float acc[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
  const int plane_o = Fo * To;
  for (int bb = b0; bb < b1; ++bb) {
    const float* g = dz + ((size_t)bb * C + c) * plane_o;
    const float* xp = x + ((size_t)bb * XC + (XC == 1 ? 0 : c)) * F * T;
    for (int e = threadIdx.x; e < plane_o; e += blockDim.x) {
      const int i = e / To, j = e - i * To;
      const float gv = g[e];
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int fi = i * STRIDE + u - P;
        const bool rok = fi >= 0 && fi < F;
#pragma unroll
        for (int v = 0; v < K; ++v) {
          const int ti = j * STRIDE + v - P;
          const float xv = (rok && ti >= 0 && ti < T) ? xp[(size_t)fi * T + ti] : 0.0f;
          acc[u * K + v] = fmaf(gv, xv, acc[u * K + v]);
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    const float t = eat::wave_sum(acc[i]);
    if (lane == 0) s_red[wv][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < K * K)
    atomicAdd(dw + ((size_t)(per_sample ? b0 * C : 0) + c) * K * K + threadIdx.x,
              s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}


// Stem weight gradient (models/mn/model.py:124-133: 3x3 / stride 2, ONE input channel, C = 16 w output channels):
// dW[c][u][v] = sum_{b,i,j} dz[b,c,i,j] x[b,0,2i+u-1,2j+v-1].  The generic kernel above walks one channel per block and
// gathers 9 predicated x values per dz element (710 us at B = 256 for 655 MB: 0.9 TB/s).  Here a thread owns output
// columns, keeps the 3x3 x window of a position in registers and applies it to 16 channels at once (16 coalesced dz
// loads per position, the window is loaded once per position and channel group), 144 accumulators per thread; one
// wave reduction + 144 atomics per block.
template <int CG>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                         float* __restrict__ dw, int C, int F, int T, int Fo, int To,
                                                         int rows_per_block) {
  __shared__ float s_red[4][CG * 9];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * rows_per_block;
  const int i1 = (i0 + rows_per_block) < Fo ? (i0 + rows_per_block) : Fo;
  const float* xb = x + (size_t)b * F * T;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += CG) {
    float acc[CG][9];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[c][t] = 0.0f;
    const float* gz = dz + ((size_t)b * C + c0) * Fo * To;
    for (int i = i0; i < i1; ++i) {
      for (int j = threadIdx.x; j < To; j += 256) {
        float xw[9];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int fi = 2 * i + u - 1;
          const bool rok = fi >= 0 && fi < F;
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            const int ti = 2 * j + v - 1;
            xw[u * 3 + v] = (rok && ti >= 0 && ti < T) ? xb[(size_t)fi * T + ti] : 0.0f;
          }
        }
        const size_t pos = (size_t)i * To + j;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          const float g = (c0 + c < C) ? gz[(size_t)c * Fo * To + pos] : 0.0f;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(g, xw[t], acc[c][t]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = eat::wave_sum(acc[c][t]);
        if (lane == 0) s_red[wv][c * 9 + t] = v;
      }
    __syncthreads();
    for (int e = threadIdx.x; e < CG * 9; e += 256) {
      const int c = e / 9;
      if (c0 + c < C) atomicAdd(dw + (size_t)(c0 + c) * 9 + (e - c * 9), s_red[0][e] + s_red[1][e] + s_red[2][e] + s_red[3][e]);
    }
    __syncthreads();
  }
}

// Column-walking variant (depthwise, XC == C): the kernel above loads K*K predicated 4-byte x values per output
// element (25 narrow loads for a 5x5) and is bound by the texture unit at ~1.4 TB/s.  Here a thread owns one output
// column of one (b, c) plane and walks down the rows with the K x K input window in a register ring, as the forward
// kernel does: K*STRIDE new x values + one dz value per output.  A block = one channel, TY samples x TX columns, and
// loops over its slice of the batch; the K*K partial sums are reduced once per block (shuffles, LDS, K*K atomics).
template <int K, int STRIDE>
__global__ __launch_bounds__(256) void dw_wgrad_col_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                           float* __restrict__ dw, int B, int C, int F, int T, int Fo,
                                                           int To, int TX, int b_per_block, int per_sample,
                                                           const float* __restrict__ in_a,
                                                           const float* __restrict__ in_b, int in_act) {
  constexpr int P = (K - 1) / 2;
  constexpr int NSLOT = K;                              // ring of K rows: step R uses slots (u + R*STRIDE) % K
  __shared__ float s_red[4][K * K];
  const int tid = threadIdx.x;
  const int tx = tid % TX, ty = tid / TX, TY = 256 / TX;
  const int to = blockIdx.x * TX + tx;
  const int c = blockIdx.y;
  const int b0 = blockIdx.z * b_per_block;
  const int b1 = (b0 + b_per_block) < B ? (b0 + b_per_block) : B;
  float acc[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
  // in_a != NULL: the conv input was act_in(in_a[c] * x + in_b[c]) evaluated on load (eat_dw_conv_fwd_tf)
  const bool has_tf = in_a != nullptr;
  const float ia = has_tf ? in_a[c] : 1.0f, ib = has_tf ? in_b[c] : 0.0f;
  if (to < To) {
    const int t0 = to * STRIDE - P;
    bool cok[K];
#pragma unroll
    for (int v = 0; v < K; ++v) cok[v] = (t0 + v >= 0) && (t0 + v < T);
    for (int bb = b0 + ty; bb < b1; bb += TY) {
      const float* g = dz + ((size_t)bb * C + c) * Fo * To + to;
      const float* xp = x + ((size_t)bb * C + c) * F * T;
      float win[NSLOT][K];
      auto load_row = [&](int fi, float (&dst)[K]) {
        const bool rok = fi >= 0 && fi < F;
        const float* src = xp + (size_t)(rok ? fi : 0) * T + t0;
        if (has_tf) {
#pragma unroll
          for (int v = 0; v < K; ++v) dst[v] = (rok && cok[v]) ? eat::activate_rt(fmaf(ia, src[v], ib), in_act) : 0.0f;
        } else {
#pragma unroll
          for (int v = 0; v < K; ++v) dst[v] = (rok && cok[v]) ? src[v] : 0.0f;
        }
      };
#pragma unroll
      for (int u = 0; u < K - STRIDE; ++u) load_row(u - P, win[u]);      // rows kept from "step -1"
      for (int fo0 = 0; fo0 < Fo; fo0 += K) {
#pragma unroll
        for (int R = 0; R < K; ++R) {                   // K steps = one full rotation of the ring
          const int fo = fo0 + R;
          if (fo < Fo) {
#pragma unroll
            for (int u = K - STRIDE; u < K; ++u) load_row(fo * STRIDE - P + u, win[(u + R * STRIDE) % NSLOT]);
            const float gv = g[(size_t)fo * To];
#pragma unroll
            for (int u = 0; u < K; ++u)
#pragma unroll
              for (int v = 0; v < K; ++v) acc[u * K + v] = fmaf(gv, win[(u + R * STRIDE) % NSLOT][v], acc[u * K + v]);
          }
        }
      }
    }
  }
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    const float t = eat::wave_sum(acc[i]);
    if (lane == 0) s_red[wv][i] = t;
  }
  __syncthreads();
  if (tid < K * K)
    atomicAdd(dw + ((size_t)(per_sample ? b0 * C : 0) + c) * K * K + tid,
              s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]);
}

// ---- pointwise weight gradient: dW[co,ci] = sum_{b,s} dz[b,co,s] x[b,ci,s] --------------------------------------
// Both operands are contiguous along the reduction axis s: each lane loads 4 consecutive s as one
// float4 and feeds them to 4 MFMAs (consistent k permutation).  Block = 4 waves on one 32 x 32
// tile of dW, each wave reducing its own slice of the (b, s) range; partials combined in LDS and
// added to dW with one atomic per element per block.
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                       const float* __restrict__ xscale, float* __restrict__ dW,
                                                       int B, int Co, int Ci, int S, int b_per_block, int per_sample,
                                                       WgTf tf, int n_slots) {
  __shared__ float s_red[3][4][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int b0 = blockIdx.z * b_per_block;
  const int b1 = (b0 + b_per_block) < B ? (b0 + b_per_block) : B;
  const int row = lane & 15, kq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool vec = (S & 3) == 0;
  auto load4 = [&](const float* base, int r, int rmax, size_t rs, int s, float (&o)[4]) {
    o[0] = o[1] = o[2] = o[3] = 0.f;
    if (r >= rmax || s >= S) return;
    const float* p = base + (size_t)r * rs + s;
    if (vec) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (s + e < S) o[e] = p[e];
    }
  };
  // x operand transform (zero padding applies to the transformed values: only loaded elements are transformed)
  const float ta0 = (tf.a && n0 + row < Ci) ? tf.a[n0 + row] : 1.0f, tb0 = (tf.a && n0 + row < Ci) ? tf.b[n0 + row] : 0.0f;
  const float ta1 = (tf.a && n0 + 16 + row < Ci) ? tf.a[n0 + 16 + row] : 1.0f, tb1 = (tf.a && n0 + 16 + row < Ci) ? tf.b[n0 + 16 + row] : 0.0f;
  auto tf4 = [&](float (&o)[4], int r, int s, float ta, float tb) {
    if (!tf.a || r >= Ci) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (s + e < S) o[e] = wg_tf(o[e], ta, tb, tf.act);
  };
  const float ac0 = (tf.actr && m0 + row < Co) ? tf.actr[m0 + row] : 0.0f;
  const float ac1 = (tf.actr && m0 + 16 + row < Co) ? tf.actr[m0 + 16 + row] : 0.0f;
  auto ctr4 = [&](float (&o)[4], int r, int s, float c) {
    if (!tf.actr || r >= Co) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (s + e < S) o[e] += c;
  };
  for (int bb = b0; bb < b1; ++bb) {
    const float* gz = dz + (size_t)bb * Co * S;
    const float* gx = x + (size_t)bb * Ci * S;
    // the conv input may be x * xscale[b, ci] (squeeze-excitation): fold the scale into the B operand
    const float sc0 = (xscale && n0 + row < Ci) ? xscale[(size_t)bb * Ci + n0 + row] : 1.0f;
    const float sc1 = (xscale && n0 + 16 + row < Ci) ? xscale[(size_t)bb * Ci + n0 + 16 + row] : 1.0f;
    for (int s0 = wv * 64; s0 < S; s0 += 256) {              // wave w takes s-blocks w, w+4, ...
      float ga[4][2][4], xb[4][2][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s0 + 16 * u + 4 * kq;
        load4(gz, m0 + row, Co, S, s, ga[u][0]);
        load4(gz, m0 + 16 + row, Co, S, s, ga[u][1]);
        load4(gx, n0 + row, Ci, S, s, xb[u][0]);
        load4(gx, n0 + 16 + row, Ci, S, s, xb[u][1]);
        tf4(xb[u][0], n0 + row, s, ta0, tb0);
        tf4(xb[u][1], n0 + 16 + row, s, ta1, tb1);
        ctr4(ga[u][0], m0 + row, s, ac0);
        ctr4(ga[u][1], m0 + 16 + row, s, ac1);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[u][i][e], xb[u][j][e] * (j ? sc1 : sc0), acc[i][j], 0, 0, 0);
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[wv - 1][i * 2 + j][r][lane] = acc[i][j][r];
  }
  __syncthreads();
  if (wv != 0) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * i + kq * 4 + r, n = n0 + 16 * j + row;   // C/D: col = lane&15, row = kq*4+r
        if (m < Co && n < Ci) {
          const int q = i * 2 + j;
          global_atomic_add(dW + (size_t)(per_sample ? (unsigned)b0 : blockIdx.z % (unsigned)(n_slots > 0 ? n_slots : 1)) * Co * Ci + (size_t)m * Ci + n,
                    acc[i][j][r] + s_red[0][q][r][lane] + s_red[1][q][r][lane] + s_red[2][q][r][lane]);
        }
      }
}


// ---- pointwise weight gradient on the bf16 matrix cores ("bf16x3", see conv_pw_bf16.hip) ---------------------------
// dW (Co x Ci) = sum over k = (b, s) of dz[co, k] x[ci, k]: both operands are contiguous along k, which is exactly the
// operand layout of v_mfma_f32_16x16x32_bf16 (lane = (row, 8 consecutive k)), so every lane loads its 8 k-values
// straight from HBM/L2 as two float4, splits them into bf16 hi + lo in registers and issues hi*hi + hi*lo + lo*hi.
// Block = 4 waves as 2 x 2 on a 128 x 128 tile of dW (each wave 64 x 64 = 4 x 4 MFMA tiles, 64 accumulator VGPRs):
// dz is read ceil(Ci/128) times and x ceil(Co/128) times (the 32 x 32-tile fp32 kernel above reads them Ci/32 and
// Co/32 times: 1.3 GB of L2 traffic for the 112 -> 672 layer instead of 0.35 GB).  The k range is cut into units of 32
// positions of one sample and split over blockIdx.z; partial tiles are added to dW with atomics.
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const bf16x2_t h = __builtin_convertvector(f32x2_t{v[i], v[i + 1]}, bf16x2_t);
    const bf16x2_t l = __builtin_convertvector(f32x2_t{v[i] - (float)h[0], v[i + 1] - (float)h[1]}, bf16x2_t);
    hi[i] = h[0]; hi[i + 1] = h[1];
    lo[i] = l[0]; lo[i + 1] = l[1];
  }
}

// LDS staging (v3): a lane-per-row direct load makes every lane of a load instruction touch a different cache line
// (measured: ~18 k cycles per 32-k step, the texture-address unit is the bottleneck).  Instead the operand tiles go
// through LDS by LDS-DMA with a coalesced mapping - lane l of one instruction fetches the 16-byte chunk
// (l & 7) ^ ((row >> 1) & 7) of row l >> 3, i.e. 8 rows x one full 128-byte line - and the XOR swizzle of the chunk
// index makes the later MFMA-fragment reads (16 lanes = 16 rows, same chunk) hit 16 different bank groups.
typedef __attribute__((address_space(3))) void wg_lds_void;
__device__ __forceinline__ void wg_glds16(const void* g, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(wg_lds_void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(g) : "memory", "m0");
}

// NPROD = 3: split operands (hi*hi + hi*lo + lo*hi, fp32-class); NPROD = 1: plain bf16 operands, fp32 accumulation
// (train_precision "bf16", BASELINE configs[2] - what autocast does to the conv weight gradient in the reference)
template <int NPROD>
__global__ __launch_bounds__(256, 2) void pw_wgrad_x3_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                             const float* __restrict__ xscale, float* __restrict__ dW,
                                                             int B, int Co, int Ci, int S, int sps, int units_per_block,
                                                             int per_sample, WgTf tf, int n_slots) {
  // stage = [A: 128 rows x 128 B][B: 128 rows x 128 B] = 32 KB; 2 stages
  __shared__ __attribute__((aligned(16))) float s_op[2][2][128 * 32];
  // wv through readfirstlane: the compiler then knows it (and the tile counts derived from it) to be wave-uniform - scalar
  // branches instead of exec-masked blocks around the loads and MFMAs of rows beyond the matrix
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 15, kg = lane >> 4;
  // (an XCD-aware workgroup order - all tiles of a k-slice on one XCD, so that their shared operand rows meet in one L2 -
  //  was measured slower in round 3 and removed in round 4; the wide-tile kernel below reads each operand once instead)
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int mb = bx * 128, nb = by * 128;                          // block tile origin
  const int mw = (wv & 1) * 64, nw = (wv >> 1) * 64;              // wave sub-tile inside the block tile
  const int total = B * sps;
  const int u0 = bz * units_per_block;
  const int u1 = (u0 + units_per_block) < total ? (u0 + units_per_block) : total;
  if (u0 >= u1) return;
  int mt_n = (Co - mb - mw + 15) / 16, nt_n = (Ci - nb - nw + 15) / 16;   // valid 16-row tiles of this wave (uniform)
  mt_n = mt_n < 0 ? 0 : (mt_n > 4 ? 4 : mt_n);
  nt_n = nt_n < 0 ? 0 : (nt_n > 4 ? 4 : nt_n);
  const bool active = mt_n > 0 && nt_n > 0;                        // idle waves still help with the loads
  // Gram matrix (dz == x, train_fuse.hip): a diagonal block's two operand tiles are the same rows - one DMA, one LDS tile
  const bool same_tile = dz == x && mb == nb && !xscale;
  // n_slots > 0: dW is a zero-filled workspace of n_slots copies, block z adds into copy z % n_slots (one block per copy
  // when n_slots == gridDim.z: bit-reproducible, reduced in a fixed order by wgrad_slot_reduce_kernel)
  const unsigned out_slot = per_sample ? (unsigned)(bz * units_per_block / sps) : (unsigned)bz % (unsigned)(n_slots > 0 ? n_slots : 1);
  const size_t out_off = (size_t)__builtin_amdgcn_readfirstlane((int)out_slot) * Co * Ci;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // loader role: this wave moves rows [32 wv, 32 wv + 32) of both operand tiles: 4 DMA instructions per operand
  const int lrow = lane >> 3;                                      // row within the 8-row group
  auto issue = [&](int bb, int stt, int stage) {
    const int s_base = stt * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 32 * wv + 8 * q + lrow;                      // row of the 128-row tile
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);             // swizzled 16-byte chunk this lane fetches
      int sidx = s_base + 4 * chunk;
      if (sidx > S - 4) sidx = S - 4;                              // tail unit: valid dummy, zeroed by the reader
      if (mb + 32 * wv + 8 * q < Co) {                             // skip row groups beyond the matrix (uniform)
        int ra = mb + row;
        if (ra >= Co) ra = Co - 1;
        wg_glds16(dz + ((size_t)bb * Co + ra) * S + sidx, &s_op[stage][0][(32 * wv + 8 * q) * 32]);
      }
      if (!same_tile && nb + 32 * wv + 8 * q < Ci) {
        int rb = nb + row;
        if (rb >= Ci) rb = Ci - 1;
        wg_glds16(x + ((size_t)bb * Ci + rb) * S + sidx, &s_op[stage][1][(32 * wv + 8 * q) * 32]);
      }
    }
  };

  int b = u0 / sps, st = u0 - b * sps;
  float sc[4] = {1.f, 1.f, 1.f, 1.f};
  float tfa[4] = {1.f, 1.f, 1.f, 1.f}, tfb[4] = {0.f, 0.f, 0.f, 0.f}, actr[4] = {0.f, 0.f, 0.f, 0.f};
  if (tf.a) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = nb + nw + 16 * j + r;
      tfa[j] = row < Ci ? tf.a[row] : 0.0f;
      tfb[j] = row < Ci ? tf.b[row] : 0.0f;
    }
  }
  if (tf.actr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = mb + mw + 16 * i + r;
      actr[i] = row < Co ? tf.actr[row] : 0.0f;
    }
  }
  int b_sc = -1;
  issue(b, st, 0);
  int stage = 0;
  for (int u = u0; u < u1; ++u) {
    int bn = b, stn = st + 1;
    if (stn == sps) { stn = 0; ++bn; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // unit u landed; the other stage is free
    if (u + 1 < u1) issue(bn, stn, stage ^ 1);
    if (active) {
      if (xscale && b != b_sc) {      // squeeze-excitation scale of the conv input, per (sample, input channel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = nb + nw + 16 * j + r;
          sc[j] = row < Ci ? xscale[(size_t)b * Ci + row] : 0.0f;
        }
        b_sc = b;
      }
      // fragment of (row, kg): global chunks 2kg and 2kg+1 of that row, stored at the swizzled slots
      const int s_lo = st * 32 + 8 * kg;
      const bool k0 = s_lo < S, k1 = s_lo + 4 < S;
      auto frag = [&](int which, int row_in_tile, bool row_ok, float scale, bf16x8_t& hi, bf16x8_t& lo,
                      bool xf = false, float fa = 1.0f, float fb = 0.0f) {
        // (indexed, not passed as a pointer: a generic pointer to LDS makes hipcc emit a flat-address check that its gfx950
        //  back end rejects - "Operand has incorrect register class")
        const int sw = (row_in_tile >> 1) & 7;
        float4 t0 = *reinterpret_cast<const float4*>(&s_op[stage][which][row_in_tile * 32 + 4 * ((2 * kg) ^ sw)]);
        float4 t1 = *reinterpret_cast<const float4*>(&s_op[stage][which][row_in_tile * 32 + 4 * ((2 * kg + 1) ^ sw)]);
        if (xf) {                                                  // block-uniform
          t0.x = wg_tf(t0.x, fa, fb, tf.act); t0.y = wg_tf(t0.y, fa, fb, tf.act); t0.z = wg_tf(t0.z, fa, fb, tf.act); t0.w = wg_tf(t0.w, fa, fb, tf.act);
          t1.x = wg_tf(t1.x, fa, fb, tf.act); t1.y = wg_tf(t1.y, fa, fb, tf.act); t1.z = wg_tf(t1.z, fa, fb, tf.act); t1.w = wg_tf(t1.w, fa, fb, tf.act);
        }
        // row_ok also guards LDS rows that were never loaded (select, not multiply: they may hold anything)
        const bool q0 = row_ok && k0, q1 = row_ok && k1;
        const float v[8] = {q0 ? t0.x * scale : 0.0f, q0 ? t0.y * scale : 0.0f, q0 ? t0.z * scale : 0.0f,
                            q0 ? t0.w * scale : 0.0f, q1 ? t1.x * scale : 0.0f, q1 ? t1.y * scale : 0.0f,
                            q1 ? t1.z * scale : 0.0f, q1 ? t1.w * scale : 0.0f};
        if constexpr (NPROD == 3) {
          split8(v, hi, lo);
        } else {
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            const bf16x2_t h = __builtin_convertvector(f32x2_t{v[i], v[i + 1]}, bf16x2_t);
            hi[i] = h[0]; hi[i + 1] = h[1];
          }
        }
      };
      bf16x8_t bh[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = nw + 16 * j + r;
        frag(same_tile ? 0 : 1, row, j < nt_n && nb + row < Ci, sc[j], bh[j], bl[j], tf.a != nullptr, tfa[j], tfb[j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < mt_n) {
          const int row = mw + 16 * i + r;
          bf16x8_t ah, al;
          frag(0, row, mb + row < Co, 1.0f, ah, al, tf.actr != nullptr, 1.0f, actr[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < nt_n) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
              if constexpr (NPROD == 3) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    b = bn; st = stn; stage ^= 1;
  }
  if (!active) return;
  float* out = dW + out_off;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!(i < mt_n && j < nt_n)) continue;                        // wave-uniform
      // a 16 x 16 tile that lies inside the matrix (wave-uniform test) adds without per-lane tests: the 64 guarded
      // atomics of a wave were 64 exec-masked blocks
      const bool full = mb + mw + 16 * i + 16 <= Co && nb + nw + 16 * j + 16 <= Ci;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = mb + mw + 16 * i + kg * 4 + q, n = nb + nw + 16 * j + r;   // C/D: row = kg*4+q, col = lane&15
        float* o = out + (size_t)m * Ci + n;
        if (full) {
          if (per_sample) *o = acc[i][j][q];
          else global_atomic_add(o, acc[i][j][q]);
        } else if (m < Co && n < Ci) {
          // per-sample gradients: the block covered the sample's whole k range - a plain store, no read-modify-write
          if (per_sample) *o = acc[i][j][q];
          else global_atomic_add(o, acc[i][j][q]);
        }
      }
    }
}

// ---- "wide-tile" weight gradient with producer / consumer waves (round 4) -------------------------------------------
// Measured on pw_wgrad_x3_kernel with its phases switched off one at a time (672 x 112 at 504 positions x 256 clips):
// loads alone 68 us (6 TB/s), fragment preparation + MFMAs alone > 100 us, the atomics 25 us - and the phases add up:
// every wave of a block goes through load wait, bf16 hi / lo split, MFMAs in lock step, so the matrix pipe idles while the
// VALUs convert and the memory pipe idles while both work.  The split itself is repeated by every wave for every fragment
// it multiplies (a row tile of the 128 x 128 block is converted by 2 waves).  And the 128-row tiles re-read the narrow
// operand once per row tile, missing L2 (PMC: FETCH_SIZE = the loads; the row tiles of a k-slice run on different XCDs):
// 694 MB loaded for 404 MB of operands.  This kernel:
//  * one block of 8 waves per CU owns a tile of up to 256 rows of the WIDE operand P (whichever of dz / x has more rows;
//    SWAP = x) times up to 160 rows of the narrow operand Q - for every mn10 layer all of Q, so both operands are read
//    once; the row tiles are balanced (672 rows = 3 tiles of 224, not 256 + 256 + 160);
//  * PRODUCER / CONSUMER waves: a workgroup's waves are dealt to the SIMDs round robin, so waves 0-3 (consumers) and 4-7
//    (producers) are one of each per SIMD.  A producer loads 1 KB pieces (8 rows x 32 positions, 16 bytes per lane: 8 full
//    128-byte lines per instruction) into REGISTERS, three 32-position units ahead (3 x 13 pieces x 4 VGPRs: the bytes in
//    flight live in the producers' otherwise idle register file, ~126 KB per CU, not in LDS), CONVERTS ONCE - 4 floats ->
//    4 bf16 hi + 4 bf16 lo with the x side's transform, SE scale and the k-tail mask applied there: 12 VALU per 4 elements
//    once instead of ~45 per fragment in each of the waves using it - and stores the fragments to one of two LDS slots.  A
//    consumer owns 64 rows of P x all of Q (up to 4 x 10 accumulator tiles) and does nothing but ds_read_b128 + MFMA.
//    One s_barrier per unit hands unit u + 1 to the consumers and the slot of unit u - 1 back to the producers: loads of
//    units u + 2 ... u + 4 and the conversion of u + 1 overlap the MFMAs of u on the same SIMD.
//  * LDS layout of a slot: row-major, 128 bytes per (row, 32 positions); the two 4-position chunks 2 kg, 2 kg + 1 of an
//    MFMA fragment share a 32-byte block [8 hi | 8 lo] (or [8 lo | 8 hi]: rows with bit 1 set swap the halves, and the block
//    index is XORed with bits 2-3 of the row, so that the 16 rows of a fragment read hit 16 different 16-byte bank groups).
using u32x2_t = __attribute__((ext_vector_type(2))) unsigned;
constexpr int WIDE_PT = 4, WIDE_QT = 10;                           // 16-row tiles per consumer wave: 64 rows of P x 160 rows of Q
constexpr int WIDE_NP = 13, WIDE_TP = 8;                            // register slots per producer wave and unit: 8 pieces of P (4 waves
                                                                   // x 8 x 8 rows = 256) and 5 of Q (160 rows)
constexpr int WIDE_RD = 3;                                         // units a producer holds in registers
template <int V> struct WideInt { static constexpr int value = V; };
// (an instantiation with the x operand read through act(a v + b) - the project conv's on-load input - made the producers the
//  pole: 40 x 120 at 2000 positions 152 us against 128 for the 128 x 128-tile kernel; those launches keep that kernel)
// RADD: an additive constant per x row (radd), applied to loaded elements only - the centred Gram matrix (same = 1)
// P16: the P operand is bf16 in HBM (act_io.h; the bf16-storage plan: P = the wide tensor - the project conv's input y_d / z_d
// when SWAP, the expand conv's gradient g otherwise - and Q the narrow fp32 one).  A producer then moves 8 bytes per lane and
// piece and its fragments need NO conversion; PTF (with SWAP, P16): the x rows are act(tf_a[ci] v + tf_b[ci]) evaluated on
// load (the on-load BatchNorm of the project conv's input), times the SE scale - unpack, 4 VALU, one v_cvt_pk per pair, which
// the fp32 instantiation could not afford next to its hi / lo split.  Host: P16 only with NPROD = 1.
template <int NPROD, bool SWAP, bool SCALE, bool RADD, bool P16 = false, bool PTF = false>
__global__ __launch_bounds__(512) void pw_wgrad_wide_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                            const float* __restrict__ xscale, float* __restrict__ dW, int B,
                                                            int Co, int Ci, int S, int sps, int units_per_block,
                                                            int p_tile_rows, int q_tile_rows,
                                                            const float* __restrict__ radd, int same,
                                                            const float* __restrict__ tf_a = nullptr,
                                                            const float* __restrict__ tf_b = nullptr, int tf_act = 0,
                                                            int ps_ns = 0) {
  // ps_ns > 0 (per-sample gradients, eat_pw_conv_dyn_wgrad_b16): gridDim.z = B * ps_ns, block z multiplies slice z % ps_ns
  // (units_per_block units, clipped to the sample) of sample z / ps_ns and stores it as copy (slice * B + sample) of dW: copy
  // 0 of every sample forms the (B, Co, Ci) result, to which the caller adds the other slices
  static_assert(!P16 || NPROD == 1, "bf16-stored operand: plain bf16 products");
  static_assert(!PTF || (P16 && SWAP), "on-load transform: the bf16-stored x operand");
  constexpr unsigned PB = P16 ? 2u : 4u;                              // bytes per element of P
  extern __shared__ __attribute__((aligned(16))) float w_smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float* __restrict__ P = SWAP ? x : dz;
  const float* __restrict__ Q = SWAP ? dz : x;
  const int PR = SWAP ? Ci : Co, QR = SWAP ? Co : Ci;
  // same (Gram matrix, dz == x, one tile): the Q operand IS the P operand - no Q pieces, Q fragments read from P's rows;
  // radd (or NULL): additive constant per x row, applied to loaded elements only (the centred Gram matrix)
  const int p0 = blockIdx.x * p_tile_rows, q0 = blockIdx.y * q_tile_rows;
  const int pv = (PR - p0) < p_tile_rows ? (PR - p0) : p_tile_rows;   // valid rows of this block's tile
  const int qv = (QR - q0) < q_tile_rows ? (QR - q0) : q_tile_rows;
  const int total = B * sps;
  int u0 = blockIdx.z * units_per_block;
  int u1 = (u0 + units_per_block) < total ? (u0 + units_per_block) : total;
  unsigned out_slot = blockIdx.z;
  if (ps_ns > 0) {
    const int sb = (int)blockIdx.z / ps_ns, sj = (int)blockIdx.z - sb * ps_ns;
    u0 = sb * sps + sj * units_per_block;
    u1 = (u0 + units_per_block) < (sb + 1) * sps ? (u0 + units_per_block) : (sb + 1) * sps;
    out_slot = (unsigned)(sj * B + sb);
  }
  if (pv <= 0 || qv <= 0 || u0 >= u1) return;                         // block-uniform, before any barrier
  const int pt = (pv + 15) >> 4, qt = (qv + 15) >> 4;                 // 16-row tiles
  const int GP = pt * 2, GQ = same ? 0 : (qv + 7) >> 3;               // 8-row pieces (P: whole 16-row tiles, so that the Q rows start
                                                                      // on a multiple of 16: one swizzle term for all tiles)
  const int GD = GP + GQ;
  const int slot_f = GD * 256;                                        // floats per LDS slot
  const int wq = wv & 3;

  // (measured without effect: s_setprio 1 / 3 for the producers - the later-dispatched half -, for the consumers, and
  //  the roles swapped: 141 - 146 us against 143 - 151 for 672 x 112)
  if (wv >= 4) {
    // ------------------------------------------------------------------ producer: loads, conversion, LDS stores
    // The loads of a unit are STRAIGHT-LINE code - every producer issues WIDE_NP of them per unit whatever the tile size
    // (slots past the last piece fetch one broadcast line), units past the block's range re-fetch the last unit, the SE scale
    // is fetched every step for the unit converted in the next one: with loads inside conditional blocks the compiler's
    // wait-count pass falls back to vmcnt(0) at every use, which drains the three units in flight.
    const int lrow = lane >> 3, chunk = lane & 7;                     // this lane's 16 bytes of a piece: row lrow, positions 4 chunk ...
    // register slot t of a producer: t < WIDE_TP -> piece wq + 4 t of P, else piece wq + 4 (t - WIDE_TP) of Q (fixed slot classes:
    // the slots that hold x rows - transform coefficients, SE scale - are known at compile time)
    constexpr int XS0 = SWAP ? 0 : WIDE_TP, XN = SWAP ? WIDE_TP : WIDE_NP - WIDE_TP;
    // per slot: byte offset of this lane's 16 bytes relative to (operand + sample offset + 32 * unit)
    unsigned roff[WIDE_NP];
    int xrow[SCALE ? XN : 1];
    float ra[RADD ? XN : 1];
    float tfa[PTF ? XN : 1], tfb[PTF ? XN : 1];
#pragma unroll
    for (int t = 0; t < WIDE_NP; ++t) {
      const bool isp = t < WIDE_TP;
      const int g = wq + 4 * (isp ? t : t - WIDE_TP);                 // piece of P / of Q (wave-uniform)
      const int mr = g * 8 + lrow, nv = isp ? pv : qv;
      const int row = (isp ? p0 : q0) + (mr > nv - 1 ? nv - 1 : mr);
      const bool valid = g < (isp ? GP : GQ);
      const unsigned eb = isp ? PB : 4u;
      roff[t] = valid ? eb * ((unsigned)row * (unsigned)S + 4u * (unsigned)chunk) : eb * (unsigned)(isp ? p0 : q0) * (unsigned)S;
      if (t >= XS0 && t < XS0 + XN) {
        const int xr = row < Ci ? row : Ci - 1;
        if constexpr (SCALE) xrow[t - XS0] = xr;
        if constexpr (RADD) ra[t - XS0] = radd[xr];
        if constexpr (PTF) { tfa[t - XS0] = tf_a[xr]; tfb[t - XS0] = tf_b[xr]; }
      }
    }
    if constexpr (PTF) {
#pragma unroll
      for (int t = 0; t < XN; ++t) asm volatile("" ::"v"(tfa[t]), "v"(tfb[t]));
    }
    // (the loads above must be back - and known to the compiler to be back - before the loop: a wait it placed at their first
    //  use inside the loop would be a vmcnt(0) per step; the empty asm statements read the registers)
    if constexpr (RADD) {
#pragma unroll
      for (int t = 0; t < XN; ++t) asm volatile("" ::"v"(ra[t]));
    }
    const bool tail = (S & 31) != 0;
    float4 buf[WIDE_RD][WIDE_NP];
    u32x2_t pbuf[WIDE_RD][P16 ? WIDE_TP : 1];                        // P16: the P slots hold 4 bf16 (not members of a float4:
                                                                      // hipcc folded `.y` of a partly written float4 into `.x`)
    float xs[SCALE ? XN : 1], xs_next[SCALE ? XN : 1];
    auto load_unit = [&](auto ktag, int bb, int stt) {
      constexpr int K = decltype(ktag)::value;
      const char* pb = reinterpret_cast<const char*>(P) + ((size_t)bb * PR * S + (size_t)stt * 32) * PB;
      const char* qb = reinterpret_cast<const char*>(Q + ((size_t)bb * QR * S + (size_t)stt * 32));
      // the sample's last unit: chunks past S fetch a valid dummy (zeroed by the converter)
      const int lim = S - 4 - stt * 32;                               // largest valid k offset inside this unit
      const unsigned back_e = (tail && stt == sps - 1 && 4 * chunk > lim) ? (unsigned)(4 * chunk - lim) : 0u;   // elements
#pragma unroll
      for (int t = 0; t < WIDE_NP; ++t) {
        const bool isp = t < WIDE_TP;
        const bool valid = wq + 4 * (isp ? t : t - WIDE_TP) < (isp ? GP : GQ);
        if (isp && P16) {                                             // (compile-time per slot) 4 bf16 = 8 bytes
          pbuf[K][t < WIDE_TP ? t : 0] = *reinterpret_cast<const u32x2_t*>(pb + (valid ? roff[t] - PB * back_e : roff[t]));
        } else {
          buf[K][t] = *reinterpret_cast<const float4*>((isp ? pb : qb) + (valid ? roff[t] - (isp ? PB : 4u) * back_e : roff[t]));
        }
      }
    };
    auto load_scale = [&](int bb) {                                   // SE scale of the conv input, per (sample, input channel)
      if constexpr (SCALE) {
#pragma unroll
        for (int t = 0; t < XN; ++t) xs_next[t] = xscale[(size_t)bb * Ci + xrow[t]];
      }
    };
    const int hb = (lrow >> 1) & 1;
    auto convert_unit = [&](auto ktag, int slot, int stt) {
      constexpr int K = decltype(ktag)::value;
      const int sb = slot * slot_f;
      const bool kz = tail && stt * 32 + 4 * chunk >= S;              // k tail (S % 4 == 0: whole chunks)
#pragma unroll
      for (int t = 0; t < WIDE_NP; ++t) {
        const bool isp = t < WIDE_TP;
        const int g = wq + 4 * (isp ? t : t - WIDE_TP);
        if (g < (isp ? GP : GQ)) {
          const int gi = isp ? g : GP + g;                            // piece of the LDS slot
          float4 w = (isp && P16) ? float4{0.f, 0.f, 0.f, 0.f} : buf[K][t];
          const int blk = (chunk >> 1) ^ (((gi & 1) << 1) | (lrow >> 2));   // 32-byte block of the row: (c >> 1) ^ ((row >> 2) & 3)
          const int rowf = sb + gi * 256 + lrow * 32 + 8 * blk + 2 * (chunk & 1);
          if (isp && P16) {                                            // (compile-time per slot) finished bf16 pairs
            unsigned w01 = pbuf[K][t < WIDE_TP ? t : 0][0], w23 = pbuf[K][t < WIDE_TP ? t : 0][1];
            if constexpr (SWAP && (SCALE || PTF)) {                    // P = the x rows: transform / SE scale on load
              float v0 = eat::bf_lo(w01), v1 = eat::bf_hi(w01), v2 = eat::bf_lo(w23), v3 = eat::bf_hi(w23);
              if constexpr (PTF) {
                const float fa = tfa[t < XN ? t : 0], fb = tfb[t < XN ? t : 0];
                v0 = wg_tf(v0, fa, fb, tf_act); v1 = wg_tf(v1, fa, fb, tf_act);
                v2 = wg_tf(v2, fa, fb, tf_act); v3 = wg_tf(v3, fa, fb, tf_act);
              }
              if constexpr (SCALE) {
                const float sc = xs[t < XN ? t : 0];
                v0 *= sc; v1 *= sc; v2 *= sc; v3 *= sc;
              }
              w01 = eat::pack_bf2(v0, v1); w23 = eat::pack_bf2(v2, v3);
            }
            if (kz) { w01 = 0u; w23 = 0u; }
            *reinterpret_cast<u32x2_t*>(&w_smem[rowf + 4 * hb]) = u32x2_t{w01, w23};
            continue;
          }
          if constexpr (SCALE) {
            if (t >= XS0 && t < XS0 + XN) {
              const float sc = xs[t - XS0 < 0 ? 0 : t - XS0];
              w.x *= sc; w.y *= sc; w.z *= sc; w.w *= sc;
            }
          }
          if constexpr (RADD) {
            if (t >= XS0 && t < XS0 + XN) {                           // centring constant of the row
              const float c = ra[t - XS0 < 0 ? 0 : t - XS0];
              w.x += c; w.y += c; w.z += c; w.w += c;
            }
          }
          if (kz) w = float4{0.f, 0.f, 0.f, 0.f};
          const bf16x2_t h01 = __builtin_convertvector(f32x2_t{w.x, w.y}, bf16x2_t);
          const bf16x2_t h23 = __builtin_convertvector(f32x2_t{w.z, w.w}, bf16x2_t);
          *reinterpret_cast<u32x2_t*>(&w_smem[rowf + 4 * hb]) = u32x2_t{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
          if constexpr (NPROD == 3) {
            const bf16x2_t l01 = __builtin_convertvector(f32x2_t{w.x - (float)h01[0], w.y - (float)h01[1]}, bf16x2_t);
            const bf16x2_t l23 = __builtin_convertvector(f32x2_t{w.z - (float)h23[0], w.w - (float)h23[1]}, bf16x2_t);
            *reinterpret_cast<u32x2_t*>(&w_smem[rowf + 4 * (1 - hb)]) = u32x2_t{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
          }
        }
      }
    };
    // (the lambdas take the unit's coordinates by value and the code below advances them: counters captured by reference
    //  and modified inside a generic lambda ended up in scratch memory)
    int bi = u0 / sps, sti = u0 - bi * sps, ui = u0;                  // next unit to load (stops at the last unit of the range)
    int bc = bi, stc = sti, uc = u0;                                  // next unit to convert
#define WIDE_LOAD(K_)                                                          \
    do {                                                                       \
      load_unit(WideInt<K_>{}, bi, sti);                                       \
      if (ui + 1 < u1) { ++ui; if (++sti == sps) { sti = 0; ++bi; } }          \
    } while (0)
    // conversion of unit uc; before it, the SE scale of the unit converted NEXT is requested (used one step later, when
    // only the loads issued after it are still in flight)
#define WIDE_CONVERT(K_, SLOT_)                                                \
    do {                                                                       \
      if constexpr (SCALE) {                                                   \
        _Pragma("unroll") for (int t = 0; t < XN; ++t) xs[t] = xs_next[t];      \
        const int bn = stc + 1 == sps ? bc + 1 : bc;                           \
        load_scale(bn < B ? bn : B - 1);                                       \
      }                                                                        \
      if (uc < u1) convert_unit(WideInt<K_>{}, SLOT_, stc);                    \
      ++uc;                                                                    \
      if (++stc == sps) { stc = 0; ++bc; }                                     \
    } while (0)
    // unit u0 + m lives in buf[m % 3] and goes to LDS slot m % 2
    load_scale(bc);
    WIDE_LOAD(0); WIDE_LOAD(1); WIDE_LOAD(2);
    WIDE_CONVERT(0, 0);
    WIDE_LOAD(0);
    // step n (after barrier n the consumers multiply unit n): convert unit n + 1, reload its registers with unit n + 4
    const int nsteps = u1 - u0;
    for (int n = 0; n < nsteps; n += 3) {
      __syncthreads();                                                // barrier n: unit n is in LDS; the slot of unit n - 1 is free
      WIDE_CONVERT(1, (n + 1) & 1);
      WIDE_LOAD(1);
      if (n + 1 >= nsteps) break;
      __syncthreads();
      WIDE_CONVERT(2, (n + 2) & 1);
      WIDE_LOAD(2);
      if (n + 2 >= nsteps) break;
      __syncthreads();
      WIDE_CONVERT(0, (n + 3) & 1);
      WIDE_LOAD(0);
    }
#undef WIDE_LOAD
#undef WIDE_CONVERT
    return;
  }

  // -------------------------------------------------------------------- consumer: fragments + MFMA
  const int r = lane & 15, kg = lane >> 4;
  const int pbase = pt >> 2, pext = pt & 3;                           // P tiles of this wave: [pm0, pm0 + pm_n)
  const int pm_n = pbase + (wq < pext ? 1 : 0), pm0 = wq * pbase + (wq < pext ? wq : pext);
  f32x4 acc[WIDE_PT][WIDE_QT];
#pragma unroll
  for (int i = 0; i < WIDE_PT; ++i)
#pragma unroll
    for (int j = 0; j < WIDE_QT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane LDS offsets (floats) of the fragment of row r (of any 16-row tile), k group kg: 16 bytes of hi, 16 bytes of lo
  // inside the row's 32-byte block kg ^ ((r >> 2) & 3), halves swapped for rows with bit 1 set (see the converter)
  const int fblk = r * 32 + 8 * (kg ^ ((r >> 2) & 3));
  const int f_hi = fblk + 4 * ((r >> 1) & 1), f_lo = fblk + 4 * (1 - ((r >> 1) & 1));
  const int p_off = pm0 * 512, q_off = same ? 0 : GP * 256;
  for (int u = u0; u < u1; ++u) {
    __syncthreads();                                                  // unit u is in LDS slot (u - u0) & 1
    if (pm_n > 0) {
      const int sb = ((u - u0) & 1) * slot_f;
      // all WIDE_PT fragments of P, whatever pm_n is: a wave with 3 tiles multiplies a 4th (the next wave's rows, or Q rows,
      // or zeros past the end of LDS) into accumulators that are never written - it would wait at the barrier for the waves
      // with 4 tiles anyway, and the loop has no data-dependent control flow around its fragment registers
      bf16x8_t ph[WIDE_PT], pl[WIDE_PT];
#pragma unroll
      for (int i = 0; i < WIDE_PT; ++i) {
        ph[i] = *reinterpret_cast<const bf16x8_t*>(&w_smem[sb + p_off + 512 * i + f_hi]);
        if constexpr (NPROD == 3) pl[i] = *reinterpret_cast<const bf16x8_t*>(&w_smem[sb + p_off + 512 * i + f_lo]);
      }
      // the Q fragment of tile j + 1 is read before the MFMAs of tile j (past the last tile it reads the other slot or zeros
      // beyond the LDS allocation - never used)
      bf16x8_t qh, ql;
      qh = *reinterpret_cast<const bf16x8_t*>(&w_smem[sb + q_off + f_hi]);
      if constexpr (NPROD == 3) ql = *reinterpret_cast<const bf16x8_t*>(&w_smem[sb + q_off + f_lo]);
#pragma unroll
      for (int j = 0; j < WIDE_QT; ++j) {
        if (j < qt) {
          bf16x8_t nh, nl;
          if (j + 1 < WIDE_QT) {
            nh = *reinterpret_cast<const bf16x8_t*>(&w_smem[sb + q_off + 512 * (j + 1) + f_hi]);
            if constexpr (NPROD == 3) nl = *reinterpret_cast<const bf16x8_t*>(&w_smem[sb + q_off + 512 * (j + 1) + f_lo]);
          }
#pragma unroll
          for (int i = 0; i < WIDE_PT; ++i) {
            // D rows (kg * 4 + e) follow the first operand, D columns (lane & 15) the second: the x / Ci index goes FIRST, so
            // that a lane's four accumulator values are four consecutive columns of dW (one 16-byte store)
            if constexpr (SWAP) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[i], qh, acc[i][j], 0, 0, 0);
              if constexpr (NPROD == 3) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[i], ql, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[i], qh, acc[i][j], 0, 0, 0);
              }
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, ph[i], acc[i][j], 0, 0, 0);
              if constexpr (NPROD == 3) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, pl[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql, ph[i], acc[i][j], 0, 0, 0);
              }
            }
          }
          if (j + 1 < WIDE_QT) { qh = nh; if constexpr (NPROD == 3) ql = nl; }
        }
      }
    }
  }
  if (pm_n <= 0) return;
  // Epilogue: PLAIN 16-byte stores into this k-slice's own copy of dW (ws holds gridDim.z copies; wgrad_slot_reduce4_kernel
  // adds them in a fixed order: bit-reproducible).  The atomic form - 160 wave-instructions of 64 fp32 atomics per consumer -
  // cost 25 - 35 us of the 85 - 145 us launches (the same-address adds of all k-slices arrive together).
  float* out = dW + (size_t)out_slot * Co * Ci;
#pragma unroll
  for (int i = 0; i < WIDE_PT; ++i)
#pragma unroll
    for (int j = 0; j < WIDE_QT; ++j) {
      if (!(i < pm_n && j < qt)) continue;                            // wave-uniform
      const int pr0 = 16 * (pm0 + i), qr0 = 16 * j;                   // tile origins inside the block tile
      // C/D layout: row = kg * 4 + e (first MFMA operand = the x rows = columns n of dW), column = lane & 15 (the dz rows m);
      // rows / columns past the matrix hold products of the clamped duplicate rows and are not written (Ci % 4 == 0)
      const int m = SWAP ? q0 + qr0 + r : p0 + pr0 + r;
      const int n = (SWAP ? p0 + pr0 : q0 + qr0) + kg * 4;
      if (m < Co && n < Ci) *reinterpret_cast<f32x4*>(out + (size_t)m * Ci + n) = acc[i][j];
    }
}

// Narrow layers (one side <= 16 channels, the other <= 64: mn10 block 1 and the expand of block 2, planes of 32000
// positions): dW is a single 64 x 64 wave tile and the gradient is a pure streaming reduction over k.  Here the direct,
// LDS-free form wins: the 4 waves of a block split the k range, every lane loads the 8 consecutive k of its row straight
// from HBM (one full 128-byte line per row and unit).  Round 3: the tile counts are compile-time (a 16 x 16 product keeps
// 8, not 128, operand registers), the loads of the NEXT unit are issued before the MFMAs of the current one (the loop
// was latency-bound: 1.7 TB/s on the 16 x 16 layers), and SAME (dz == x: the Gram matrix of train_fuse.hip) loads once.
template <int MTN, int NTN, bool SAME>
__global__ __launch_bounds__(256, 2) void pw_wgrad_x3_narrow_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                    const float* __restrict__ xscale, float* __restrict__ dW,
                                                                    int B, int Co, int Ci, int S, int sps,
                                                                    int units_per_block, int n_slots, WgTf tf, int ps_spl) {
  // the four waves' tiles are combined in LDS (ds_add_f32) before ONE set of global atomics per block, and the blocks
  // are spread over n_slots copies of dW (reduced by wgrad_slot_reduce_kernel): atomics on the same address serialise
  // in L2 at ~40 ns each, which made the 8192 (= 2048 blocks x 4 waves) adds per element the whole cost of this kernel
  // on the 16 x 16 layers (311 us for 105 us of HBM time)
  __shared__ float s_tile[MTN * NTN * 256];
  // wv through readfirstlane: the compiler then knows it (and the tile counts derived from it) to be wave-uniform - scalar
  // branches instead of exec-masked blocks around the loads and MFMAs of rows beyond the matrix
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 15, kg = lane >> 4;
  // round 3: blockIdx.x / blockIdx.y select a group of MTN / NTN row tiles ("thin" matrices: up to ~8 x 8 tiles over a
  // long k axis stream faster through this LDS-free kernel than through the barrier-per-32-positions LDS pipeline)
  const int m0 = blockIdx.x * (16 * MTN), n0 = SAME ? m0 : blockIdx.y * (16 * NTN);
  // ps_spl > 0: per-sample gradients (DyMN, models/dymn/dy_block.py:120-127 backward): ps_spl blocks share a sample's k
  // range and add into that sample's own Co x Ci matrix
  const int ps_b = ps_spl > 0 ? (int)blockIdx.z / ps_spl : 0;
  const int total = ps_spl > 0 ? (ps_b + 1) * sps : B * sps;
  const int u0 = ps_spl > 0 ? ps_b * sps + ((int)blockIdx.z - ps_b * ps_spl) * units_per_block : (int)blockIdx.z * units_per_block;
  const int u1 = (u0 + units_per_block) < total ? (u0 + units_per_block) : total;
  for (int i = threadIdx.x; i < MTN * NTN * 256; i += 256) s_tile[i] = 0.0f;
  __syncthreads();
  f32x4 acc[MTN][NTN];
#pragma unroll
  for (int i = 0; i < MTN; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sc[NTN], tfa[NTN], tfb[NTN], actr[MTN];
#pragma unroll
  for (int j = 0; j < NTN; ++j) {
    sc[j] = 1.0f;
    const int row = n0 + 16 * j + r;
    tfa[j] = (tf.a && row < Ci) ? tf.a[row] : 1.0f;
    tfb[j] = (tf.a && row < Ci) ? tf.b[row] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < MTN; ++i) {
    const int row = m0 + 16 * i + r;
    actr[i] = (tf.actr && row < Co) ? tf.actr[row] : 0.0f;
  }
  int b_sc = -1;

  auto load = [&](int u, float (&av)[MTN][8], float (&bv)[SAME ? 1 : NTN][8]) {
    const bool live = u < u1;                                   // wave-uniform
    const int uu = live ? u : u0;
    const int b = uu / sps, st = uu - b * sps;
    const int s = st * 32 + 8 * kg;
#pragma unroll
    for (int i = 0; i < MTN; ++i) {
      const int row = m0 + 16 * i + r;
      const float* p = dz + ((size_t)b * Co + (row < Co ? row : Co - 1)) * S + s;
      const bool ok0 = live && row < Co && s < S, ok1 = ok0 && s + 4 < S;
      const float4 t0 = ok0 ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 t1 = ok1 ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float c0 = ok0 ? actr[i] : 0.0f, c1 = ok1 ? actr[i] : 0.0f;      // (0 without centring: elements unchanged)
      av[i][0] = t0.x + c0; av[i][1] = t0.y + c0; av[i][2] = t0.z + c0; av[i][3] = t0.w + c0;
      av[i][4] = t1.x + c1; av[i][5] = t1.y + c1; av[i][6] = t1.z + c1; av[i][7] = t1.w + c1;
    }
    if constexpr (!SAME) {
#pragma unroll
      for (int j = 0; j < NTN; ++j) {
        const int row = n0 + 16 * j + r;
        const float* p = x + ((size_t)b * Ci + (row < Ci ? row : Ci - 1)) * S + s;
        const bool ok0 = live && row < Ci && s < S, ok1 = ok0 && s + 4 < S;
        const float4 t0 = ok0 ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 t1 = ok1 ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[j][0] = t0.x; bv[j][1] = t0.y; bv[j][2] = t0.z; bv[j][3] = t0.w;
        bv[j][4] = t1.x; bv[j][5] = t1.y; bv[j][6] = t1.z; bv[j][7] = t1.w;
        if (tf.a) {                                               // block-uniform; invalid elements stay 0
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[j][e] = (e < 4 ? ok0 : ok1) ? wg_tf(bv[j][e], tfa[j], tfb[j], tf.act) : 0.0f;
        }
      }
    }
  };
  auto compute = [&](int u, const float (&av)[MTN][8], const float (&bv)[SAME ? 1 : NTN][8]) {
    if (u >= u1) return;                                        // wave-uniform
    if (!SAME && xscale) {
      const int b = u / sps;
      if (b != b_sc) {
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
          const int row = n0 + 16 * j + r;
          sc[j] = row < Ci ? xscale[(size_t)b * Ci + row] : 0.0f;
        }
        b_sc = b;
      }
    }
    bf16x8_t ah[MTN], al[MTN];
#pragma unroll
    for (int i = 0; i < MTN; ++i) split8(av[i], ah[i], al[i]);
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
      bf16x8_t bh, bl;
      if constexpr (SAME) {
        bh = ah[j]; bl = al[j];                                 // host: MTN == NTN
      } else {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = bv[j][e] * sc[j];
        split8(t, bh, bl);
      }
#pragma unroll
      for (int i = 0; i < MTN; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
      }
    }
  };

  float a0[MTN][8], a1[MTN][8], b0[SAME ? 1 : NTN][8], b1[SAME ? 1 : NTN][8];
  load(u0 + wv, a0, b0);
  for (int u = u0 + wv; u < u1; u += 8) {
    load(u + 4, a1, b1);
    compute(u, a0, b0);
    load(u + 8, a0, b0);
    compute(u + 4, a1, b1);
  }
  // the four waves add their tiles in a FIXED order (wave 0, 1, 2, 3): with one slot per block the result is then
  // bit-reproducible from run to run - needed for the Gram matrix, whose round-off reaches the BatchNorm statistics
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int i = 0; i < MTN; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) s_tile[((i * NTN + j) * 4 + q) * 64 + lane] += acc[i][j][q];
    }
    __syncthreads();
  }
  float* out = dW + (size_t)(ps_spl > 0 ? ps_b : (int)(blockIdx.z % n_slots)) * Co * Ci;
  for (int e = threadIdx.x; e < MTN * NTN * 256; e += 256) {
    const int ln = e & 63, q = (e >> 6) & 3, ij = e >> 8;
    const int i = ij / NTN, j = ij - i * NTN;
    const int m = m0 + 16 * i + (ln >> 4) * 4 + q, n = n0 + 16 * j + (ln & 15);
    if (m < Co && n < Ci) atomicAdd(out + (size_t)m * Ci + n, s_tile[e]);
  }
}

// dW[e] += sum over the slots of ws, in a fixed order (bit-reproducible): a block owns 64 consecutive elements, its four
// waves split the slots (wave w: slots w, w+4, ...; the loads of 16 slots are in flight together), the four partial
// sums are added in wave order.  (One thread per element walking up to 1024 slots took ~250 us: a serial latency chain.)
__global__ __launch_bounds__(256) void wgrad_slot_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int n,
                                                                int n_slots) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  float t = 0.0f;
  if (e < n) {
    int sidx = wv;
    for (; sidx + 60 < n_slots; sidx += 64) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = ws[(size_t)(sidx + 4 * q) * n + e];
#pragma unroll
      for (int q = 0; q < 16; ++q) t += v[q];
    }
    for (; sidx < n_slots; sidx += 4) t += ws[(size_t)sidx * n + e];
  }
  s_part[wv][lane] = t;
  __syncthreads();
  if (wv == 0 && e < n) dW[e] += ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
}

// the same for n % 4 == 0 with 16-byte loads: a block adds 256 elements of up to n_slots copies (the wide-tile kernel's
// per-k-slice copies: 25 - 39 MB per launch on the late mn10 layers)
__global__ __launch_bounds__(256) void wgrad_slot_reduce4_kernel(const float* __restrict__ ws, float* __restrict__ dW, int n,
                                                                 int n_slots) {
  __shared__ f32x4 s_part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = blockIdx.x * 256 + 4 * lane;
  f32x4 t{0.f, 0.f, 0.f, 0.f};
  if (e < n) {
    int sidx = wv;
    for (; sidx + 28 < n_slots; sidx += 32) {
      f32x4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(ws + (size_t)(sidx + 4 * q) * n + e);
#pragma unroll
      for (int q = 0; q < 8; ++q) t += v[q];
    }
    for (; sidx < n_slots; sidx += 4) t += *reinterpret_cast<const f32x4*>(ws + (size_t)sidx * n + e);
  }
  s_part[wv][lane] = t;
  __syncthreads();
  if (wv == 0 && e < n) {
    f32x4* o = reinterpret_cast<f32x4*>(dW + e);
    *o = *o + (((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane]);
  }
}

template <int MTN, int NTN>
static void launch_narrow(const float* dz, const float* x, const float* x_scale, float* dW, int B, int Co, int Ci, int S,
                          int sps, int upb, unsigned nz, int n_slots, hipStream_t s, WgTf tf, int mg, int ng, bool gram,
                          int ps_spl = 0) {
  if (gram && MTN == NTN)
    hipLaunchKernelGGL((pw_wgrad_x3_narrow_kernel<MTN, (MTN == NTN ? NTN : 1), (MTN == NTN)>), dim3(1, 1, nz), dim3(256), 0, s, dz,
                       x, x_scale, dW, B, Co, Ci, S, sps, upb, n_slots, tf, 0);
  else
    hipLaunchKernelGGL((pw_wgrad_x3_narrow_kernel<MTN, NTN, false>), dim3(mg, ng, nz), dim3(256), 0, s, dz, x, x_scale, dW, B,
                       Co, Ci, S, sps, upb, n_slots, tf, ps_spl);
}

}  // namespace

#define EAT_PLANES_GRID(B, C) dim3((unsigned)((B) * (C)))

// samples per block of the small-plane reducers: ~2048 blocks; 0 = use the one-block-per-plane kernels
static int bn_multi_ppb(int B, int C, int S) {
  if ((S & 3) != 0 || S > 2048 || (long long)B * C <= 4096) return 0;
  long long ppb = ((long long)B * C + 2047) / 2048;
  return (int)(ppb > B ? B : ppb);
}

extern "C" int eat_bn_stats(const float* z, int B, int C, int S, double* sums, eat_stream_t stream) {
  eat::clear_stale_error();
  if (const int ppb = bn_multi_ppb(B, C, S)) {
    hipLaunchKernelGGL(bn_stats_multi_kernel, dim3(C, (B + ppb - 1) / ppb), dim3(256), 0, (hipStream_t)stream, z, B, C, S >> 2,
                       ppb, sums);
    return eat::check_launch("eat_bn_stats");
  }
  hipLaunchKernelGGL(bn_stats_kernel, EAT_PLANES_GRID(B, C), dim3(S >= 1024 ? 256 : 64), 0, (hipStream_t)stream, z, C, S,
                     sums);
  return eat::check_launch("eat_bn_stats");
}

extern "C" int eat_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, double n, int C, float* a, float* b,
                               float* mean, float* invstd, eat_stream_t stream) {
  eat::clear_stale_error();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums, gamma, beta,
                     running_mean, running_var, momentum, eps, n, C, a, b, mean, invstd);
  return eat::check_launch("eat_bn_finalize");
}

extern "C" int eat_bn_act_fwd(const float* z, const float* a, const float* b, const float* res, float* y,
                              float* pool, int B, int C, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_bn_act_fwd: bad act %d", act);
  const dim3 blk(S >= 1024 ? 256 : 64);
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_fwd_kernel<ACT>), EAT_PLANES_GRID(B, C), blk, 0, (hipStream_t)stream, z,
                                           a, b, res, y, pool, C, S));
  return eat::check_launch("eat_bn_act_fwd");
}

extern "C" int eat_bn_act_bwd_reduce(const float* dy, const float* z, const float* a, const float* b,
                                     const float* mean, const float* invstd, const float* gscale, const float* gadd,
                                     int B, int C, int S, int act, double* sums, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_bn_act_bwd_reduce: bad act %d", act);
  if (const int ppb = bn_multi_ppb(B, C, S)) {
    EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_bwd_reduce_multi_kernel<ACT>), dim3(C, (B + ppb - 1) / ppb), dim3(256), 0,
                                             (hipStream_t)stream, dy, z, a, b, mean, invstd, gscale, gadd, B, C, S >> 2, ppb, sums));
    return eat::check_launch("eat_bn_act_bwd_reduce");
  }
  const dim3 blk(S >= 1024 ? 256 : 64);
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<ACT>), EAT_PLANES_GRID(B, C), blk, 0,
                                           (hipStream_t)stream, dy, z, a, b, mean, invstd, gscale, gadd, C, S, sums));
  return eat::check_launch("eat_bn_act_bwd_reduce");
}

// ---- bf16 copy of a NARROW fp32 tensor of the bf16-storage plan (block input / project-BatchNorm gradient of the widest
// blocks): the 1x1 conv kernel rounds its fp32 operand to bf16 in any case (same RNE rounding: the conv results are
// bit-identical), but it stages a bf16 operand at half the L2 -> LDS traffic and with two LDS stages - on the 448 -> 2688
// expand conv at S = 504, B = 128 that is 324 -> 238 us for a 24 us copy.
namespace {
__global__ __launch_bounds__(256) void cast_b16_kernel(const float* __restrict__ x, eat::bf16_t* __restrict__ y, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const float4 p = reinterpret_cast<const float4*>(x)[2 * i], q = reinterpret_cast<const float4*>(x)[2 * i + 1];
    uint4 o;
    o.x = eat::pack_bf2(p.x, p.y); o.y = eat::pack_bf2(p.z, p.w); o.z = eat::pack_bf2(q.x, q.y); o.w = eat::pack_bf2(q.z, q.w);
    reinterpret_cast<uint4*>(y)[i] = o;
  }
}
}  // namespace
extern "C" int eat_cast_b16(const float* x, void* y, long long n, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !y || n < 8 || (n & 7)) return eat::fail(EAT_EINVAL, "eat_cast_b16: n=%lld must be a positive multiple of 8", n);
  const long long n8 = n >> 3;
  const long long blocks = (n8 + 255) / 256;
  hipLaunchKernelGGL(cast_b16_kernel, dim3((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                     x, reinterpret_cast<eat::bf16_t*>(y), n8);
  return eat::check_launch("eat_cast_b16");
}

// ---- the two stand-alone BatchNorm passes of the bf16-storage plan (act_io.h; BASELINE configs[2]): the depthwise output
// z_d and the gradient arriving at it are bf16 in HBM.  Same arithmetic as the fp32 entry points; y (or NULL) is written in
// bf16 and `pool` sums the ROUNDED values - what the project conv will read.  (S % 4 != 0: element-wise path.)
extern "C" int eat_bn_act_fwd_b16(const void* z, const float* a, const float* b, const float* res, void* y, int y_b16,
                                  void* y_copy16, float* pool, int B, int C, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_bn_act_fwd_b16: bad act %d", act);
  if (!z || B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_bn_act_fwd_b16: bad shape");
  if ((res || y_copy16) && y_b16)
    return eat::fail(EAT_EINVAL, "eat_bn_act_fwd_b16: a residual / a bf16 copy goes with an fp32 output only");
  const dim3 blk(S >= 1024 ? 256 : 64);
  const eat::bf16_t* z16 = reinterpret_cast<const eat::bf16_t*>(z);
  if (y_b16)
    EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_fwd_kernel<ACT, eat::bf16_t>), EAT_PLANES_GRID(B, C), blk, 0, (hipStream_t)stream,
                                             z16, a, b, (const float*)nullptr, reinterpret_cast<eat::bf16_t*>(y), pool, C, S));
  else
    EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_fwd_kernel<ACT, eat::bf16_t, float>), EAT_PLANES_GRID(B, C), blk, 0,
                                             (hipStream_t)stream, z16, a, b, res, reinterpret_cast<float*>(y), pool, C, S,
                                             reinterpret_cast<eat::bf16_t*>(y_copy16)));
  return eat::check_launch("eat_bn_act_fwd_b16");
}

extern "C" int eat_bn_act_bwd_reduce_b16(const void* dy, int dy_b16, const void* z, const float* a, const float* b,
                                         const float* mean, const float* invstd, const float* gscale, const float* gadd, int B,
                                         int C, int S, int act, double* sums, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_bn_act_bwd_reduce_b16: bad act %d", act);
  if (!dy || !z || B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_bn_act_bwd_reduce_b16: bad shape");
  const eat::bf16_t* z16 = reinterpret_cast<const eat::bf16_t*>(z);
  const int ppb = bn_multi_ppb(B, C, S);
  const dim3 blk(S >= 1024 ? 256 : 64);
#define EAT_RED16(DT_, dyp)                                                                                                  \
  do {                                                                                                                      \
    if (ppb) {                                                                                                              \
      EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_bwd_reduce_multi_kernel<ACT, eat::bf16_t, DT_>), dim3(C, (B + ppb - 1) / ppb), \
                                               dim3(256), 0, (hipStream_t)stream, dyp, z16, a, b, mean, invstd, gscale, gadd, B, C, \
                                               S >> 2, ppb, sums));                                                         \
    } else {                                                                                                                \
      EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<ACT, eat::bf16_t, DT_>), EAT_PLANES_GRID(B, C), blk, 0,   \
                                               (hipStream_t)stream, dyp, z16, a, b, mean, invstd, gscale, gadd, C, S, sums)); \
    }                                                                                                                       \
  } while (0)
  if (dy_b16) EAT_RED16(eat::bf16_t, reinterpret_cast<const eat::bf16_t*>(dy));
  else EAT_RED16(float, reinterpret_cast<const float*>(dy));
#undef EAT_RED16
  return eat::check_launch("eat_bn_act_bwd_reduce_b16");
}

// apply pass over a bf16-stored z (the project conv's output z_p in the bf16-storage plan): dy and dz are fp32
extern "C" int eat_bn_act_bwd_apply_b16(const float* dy, const void* z, const float* a, const float* b, const float* mean,
                                        const float* invstd, const float* gscale, const float* gadd, const double* sums,
                                        float* dz, void* dz_copy16, int B, int C, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_bn_act_bwd_apply_b16: bad act %d", act);
  if (!dy || !z || !dz || B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_bn_act_bwd_apply_b16: bad shape");
  const dim3 blk(S >= 1024 ? 256 : 64);
  const double n = (double)B * S;
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<ACT, eat::bf16_t>), EAT_PLANES_GRID(B, C), blk, 0,
                                           (hipStream_t)stream, dy, reinterpret_cast<const eat::bf16_t*>(z), a, b, mean, invstd,
                                           gscale, gadd, sums, dz, C, S, n, reinterpret_cast<eat::bf16_t*>(dz_copy16)));
  return eat::check_launch("eat_bn_act_bwd_apply_b16");
}

// ... with dy AND dz in bf16 too (dz may alias dy): the expand BatchNorm of a DyMN block under the bf16-storage plan - g_e, z_e
// and dz_e are all wide tensors (models/dymn/dy_block.py:313-318 backward)
extern "C" int eat_bn_bwd_apply_b16(const void* dy, const void* z, const float* a, const float* b, const float* mean,
                                    const float* invstd, const double* sums, void* dz, int B, int C, int S, int act,
                                    eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_bn_bwd_apply_b16: bad act %d", act);
  if (!dy || !z || !dz || !sums || B < 1 || C < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_bn_bwd_apply_b16: bad arguments");
  const dim3 blk(S >= 1024 ? 256 : 64);
  const double n = (double)B * S;
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<ACT, eat::bf16_t, eat::bf16_t>), EAT_PLANES_GRID(B, C), blk, 0,
                                           (hipStream_t)stream, reinterpret_cast<const eat::bf16_t*>(dy),
                                           reinterpret_cast<const eat::bf16_t*>(z), a, b, mean, invstd, (const float*)nullptr,
                                           (const float*)nullptr, sums, reinterpret_cast<eat::bf16_t*>(dz), C, S, n,
                                           (eat::bf16_t*)nullptr));
  return eat::check_launch("eat_bn_bwd_apply_b16");
}

extern "C" int eat_bn_act_bwd_apply(const float* dy, const float* z, const float* a, const float* b,
                                    const float* mean, const float* invstd, const float* gscale, const float* gadd,
                                    const double* sums, float* dz, int B, int C, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_bn_act_bwd_apply: bad act %d", act);
  const dim3 blk(S >= 1024 ? 256 : 64);
  const double n = (double)B * S;
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<ACT>), EAT_PLANES_GRID(B, C), blk, 0,
                                           (hipStream_t)stream, dy, z, a, b, mean, invstd, gscale, gadd, sums, dz, C, S, n));
  return eat::check_launch("eat_bn_act_bwd_apply");
}

extern "C" int eat_plane_dot(const float* u, const float* v, const float* a, const float* b, float* out, int B,
                             int C, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_plane_dot: bad act %d", act);
  const dim3 blk(S >= 1024 ? 256 : 64);
  EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((plane_dot_kernel<ACT>), EAT_PLANES_GRID(B, C), blk, 0, (hipStream_t)stream, u,
                                           v, a, b, out, C, S));
  return eat::check_launch("eat_plane_dot");
}

static int dw_dgrad_impl(const float* dz, const float* w, const float* res, float* dx, int B, int C, int F, int T,
                         int Fo, int To, int k, int stride, int per_plane_w, eat_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (stride == 1 && (k == 3 || k == 5)) {
    return eat::dw_conv_dgrad_s1(dz, w, nullptr, res, dx, B, C, F, T, k, per_plane_w, s);
  }
  if (stride == 2 && (k == 3 || k == 5)) {
    // tile kernel (dw_plane.hip): one dz column per lane, dx row segments as 8-byte stores; 1 = not applicable
    const int rc = eat::dw_tile_dgrad2_try(dz, w, res, dx, B, C, F, T, Fo, To, k, per_plane_w, s);
    if (rc != 1) return rc;
    dim3 g2((T + 63) / 64, (B * C + 3) / 4);
    if (k == 3) hipLaunchKernelGGL((dw_dgrad_s2_kernel<3>), g2, dim3(256), 0, s, dz, w, res, dx, B * C, C, F, T, Fo, To, per_plane_w);
    else hipLaunchKernelGGL((dw_dgrad_s2_kernel<5>), g2, dim3(256), 0, s, dz, w, res, dx, B * C, C, F, T, Fo, To, per_plane_w);
    return eat::check_launch("eat_dw_conv_dgrad");
  }
  int gx = (F * T + 255) / 256;
  if (gx > 64) gx = 64;
  dim3 grid(gx, B * C);
#define EAT_DG(KK, SS) hipLaunchKernelGGL((dw_dgrad_kernel<KK, SS>), grid, dim3(256), 0, s, dz, w, res, dx, C, F, T, Fo, To, per_plane_w)
  if (k == 3 && stride == 1) EAT_DG(3, 1);
  else if (k == 3 && stride == 2) EAT_DG(3, 2);
  else if (k == 5 && stride == 1) EAT_DG(5, 1);
  else if (k == 5 && stride == 2) EAT_DG(5, 2);
  else return eat::fail(EAT_EINVAL, "eat_dw_conv_dgrad: unsupported k=%d stride=%d", k, stride);
#undef EAT_DG
  return eat::check_launch("eat_dw_conv_dgrad");
}

extern "C" int eat_dw_conv_dgrad(const float* dz, const float* w, const float* res, float* dx, int B, int C, int F,
                                 int T, int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  return dw_dgrad_impl(dz, w, res, dx, B, C, F, T, Fo, To, k, stride, 0, stream);
}

// Depthwise data gradient with the backward of the PRECEDING (forward order) BatchNorm + activation started in its
// epilogue: g = dgrad(dz) * act'(ga[c] * gz + gb[c]) and the per-wave partial sums of g (gpart [b][C][inner]); gz is the
// pre-BN output of the expand conv (same shape as g).  See train_fuse.hip for what consumes g / gpart.
extern "C" int eat_dw_conv_dgrad_g(const float* dz, const float* w, const float* gz, const float* ga, const float* gb,
                                   int gact, float* g, float* gpart, int inner_cap, int* h_inner, int B, int C, int F,
                                   int T, int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!gz || !ga || !gb || !gpart || !h_inner) return eat::fail(EAT_EINVAL, "eat_dw_conv_dgrad_g: gz, ga, gb, gpart, h_inner are required");
  if (gact < 0 || gact > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_dgrad_g: bad act %d", gact);
  if (inner_cap < eat_dw_partials_inner(F, T, Fo, To, k, stride, 1))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dgrad_g: partial buffer too small (inner_cap %d)", inner_cap);
  hipStream_t s = (hipStream_t)stream;
  if ((k == 3 || k == 5) && (stride == 1 || stride == 2)) {
    int inner = 1;
    const eat::DwEpi epi{nullptr, gz, ga, gb, gact, gpart, &inner};
    const int rc = stride == 1 ? eat::dw_conv_dgrad_s1(dz, w, nullptr, nullptr, g, B, C, F, T, k, 0, s, &epi)
                               : eat::dw_tile_dgrad2_try(dz, w, nullptr, g, B, C, F, T, Fo, To, k, 0, s, &epi);
    if (rc != 1) { *h_inner = inner; return rc; }
  }
  const int rc = dw_dgrad_impl(dz, w, nullptr, g, B, C, F, T, Fo, To, k, stride, 0, stream);
  if (rc != 0) return rc;
  *h_inner = 1;
  return eat::act_grad_sum(g, gz, ga, gb, gact, g, gpart, B, C, F * T, s);
}

extern "C" int eat_dw_conv_dyn_dgrad(const float* dz, const float* w_bc, const float* res, float* dx, int B, int C,
                                     int F, int T, int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  return dw_dgrad_impl(dz, w_bc, res, dx, B, C, F, T, Fo, To, k, stride, 1, stream);
}

static int dw_wgrad_impl(const float* dz, const float* x, float* dw, int B, int C, int XC, int F, int T, int Fo, int To,
                         int k, int stride, int per_sample, eat_stream_t stream, const float* in_a = nullptr,
                         const float* in_b = nullptr, int in_act = 0) {
  if (XC != C && XC != 1) return eat::fail(EAT_EINVAL, "eat_dw_conv_wgrad: x must have C or 1 channels");
  if (in_a && XC != C) return eat::fail(EAT_EINVAL, "eat_dw_conv_wgrad_tf: needs the column-walking kernel");
  if (XC == C) {
    // register-resident kernels (dw_plane.hip): every element loaded once; 1 = geometry not instantiated
    const int rc = eat::dw_plane_wgrad_try(dz, x, dw, B, C, F, T, Fo, To, k, stride, per_sample, in_a, in_b, in_act,
                                           (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  if (XC == C && !per_sample && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
    // column-walking kernel: block = (column tile, channel, batch slice)
    const int TX = To > 32 ? 64 : 32, TY = 256 / TX;
    const int ct = (To + TX - 1) / TX;
    int bpb = per_sample ? 1 : B;
    if (!per_sample) {
      // ~2048 blocks, but every thread should walk several planes before the block-wide reduction
      long long want = (2048 + (long long)C * ct - 1) / ((long long)C * ct);
      if (want < 1) want = 1;
      bpb = (int)((B + want - 1) / want);
      if (bpb < 4 * TY) bpb = 4 * TY < B ? 4 * TY : B;
    }
    dim3 grid(ct, C, (B + bpb - 1) / bpb);
    hipStream_t s = (hipStream_t)stream;
#define EAT_WGC(KK, SS) hipLaunchKernelGGL((dw_wgrad_col_kernel<KK, SS>), grid, dim3(256), 0, s, dz, x, dw, B, C, F, T, Fo, To, TX, bpb, per_sample, in_a, in_b, in_act)
    if (k == 3 && stride == 1) EAT_WGC(3, 1);
    else if (k == 3 && stride == 2) EAT_WGC(3, 2);
    else if (k == 5 && stride == 1) EAT_WGC(5, 1);
    else EAT_WGC(5, 2);
#undef EAT_WGC
    return eat::check_launch("eat_dw_conv_wgrad");
  }
  if (XC == 1 && k == 3 && stride == 2 && !per_sample) {
    // ~2048 blocks: (row chunks) x (samples)
    int rpb = (int)(((long long)Fo * B + 2047) / 2048);
    if (rpb < 1) rpb = 1;
    hipLaunchKernelGGL(stem_wgrad_kernel<16>, dim3((Fo + rpb - 1) / rpb, B), dim3(256), 0, (hipStream_t)stream, dz, x, dw, C, F, T,
                       Fo, To, rpb);
    return eat::check_launch("eat_dw_conv_wgrad(stem)");
  }
  // enough blocks to fill the chip: split the batch when there are few channels
  int splits = (2048 + C - 1) / C;
  if (splits > B || per_sample) splits = B;
  const int bpb = (B + splits - 1) / splits;
  dim3 grid(C, (B + bpb - 1) / bpb);
  hipStream_t s = (hipStream_t)stream;
#define EAT_WG(KK, SS) hipLaunchKernelGGL((dw_wgrad_kernel<KK, SS>), grid, dim3(256), 0, s, dz, x, dw, B, C, XC, F, T, Fo, To, bpb, per_sample)
  if (k == 3 && stride == 1) EAT_WG(3, 1);
  else if (k == 3 && stride == 2) EAT_WG(3, 2);
  else if (k == 5 && stride == 1) EAT_WG(5, 1);
  else if (k == 5 && stride == 2) EAT_WG(5, 2);
  else return eat::fail(EAT_EINVAL, "eat_dw_conv_wgrad: unsupported k=%d stride=%d", k, stride);
#undef EAT_WG
  return eat::check_launch("eat_dw_conv_wgrad");
}

extern "C" int eat_dw_conv_wgrad(const float* dz, const float* x, float* dw, int B, int C, int XC, int F, int T,
                                 int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  return dw_wgrad_impl(dz, x, dw, B, C, XC, F, T, Fo, To, k, stride, 0, stream);
}

extern "C" int eat_dw_conv_wgrad_tf(const float* dz, const float* x, const float* in_a, const float* in_b, int in_act,
                                    float* dw, int B, int C, int F, int T, int Fo, int To, int k, int stride,
                                    eat_stream_t stream) {
  eat::clear_stale_error();
  if (!in_a || !in_b) return eat::fail(EAT_EINVAL, "eat_dw_conv_wgrad_tf: in_a and in_b are required");
  if (in_act < 0 || in_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_wgrad_tf: bad in_act %d", in_act);
  return dw_wgrad_impl(dz, x, dw, B, C, C, F, T, Fo, To, k, stride, 0, stream, in_a, in_b, in_act);
}

// Backward of the depthwise conv of an inverted-residual block in ONE pass (autograd of models/mn/block_types.py:150-162
// + the first half of the backward of the expand conv's BatchNorm + activation): from dz (B,C,Fo,To) and the pre-BN expand
// output x (B,C,F,T) with its BN affine (in_a, in_b) and activation in_act
//   dw (C,k,k) += weight gradient w.r.t. the conv input act(in_a x + in_b)            [dw zeroed by the caller]
//   g (B,C,F,T) = dgrad(dz) * act'(in_a x + in_b),   gpart [B][C][inner] = per-tile sums of g
// = eat_dw_conv_wgrad_tf + eat_dw_conv_dgrad_g with dz and x read once.  inner_cap >= eat_dw_bwd_partials_inner(...)
// AND >= eat_dw_partials_inner(..., 1) (the two-kernel fallback writes its own layout); *h_inner receives inner.
extern "C" int eat_dw_conv_bwd_g(const float* dz, const float* x, const float* in_a, const float* in_b, int in_act,
                                 const float* w, float* g, float* dw, float* gpart, int inner_cap, int* h_inner, int B,
                                 int C, int F, int T, int Fo, int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!in_a || !in_b || !gpart || !h_inner) return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_g: in_a, in_b, gpart, h_inner are required");
  if (in_act < 0 || in_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_g: bad act %d", in_act);
  if (inner_cap < eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride) || inner_cap < eat_dw_partials_inner(F, T, Fo, To, k, stride, 1))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_g: partial buffer too small (inner_cap %d)", inner_cap);
  if ((k == 3 || k == 5) && (stride == 1 || stride == 2)) {
    const int rc = eat::dw_bwd_try(dz, x, in_a, in_b, in_act, w, g, dw, gpart, h_inner, B, C, F, T, Fo, To, k, stride,
                                   (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  int rc = dw_wgrad_impl(dz, x, dw, B, C, C, F, T, Fo, To, k, stride, 0, stream, in_a, in_b, in_act);
  if (rc != 0) return rc;
  return eat_dw_conv_dgrad_g(dz, w, x, in_a, in_b, in_act, g, gpart, inner_cap, h_inner, B, C, F, T, Fo, To, k, stride, stream);
}

// The same with the BatchNorm + activation backward of THIS conv's output evaluated on load (dw_plane.hip, DzBn): dy is the
// gradient w.r.t. act(BN(z)) (times gscale[b,c] plus gadd[b,c] for a squeeze-excitation block), sums the fp64 channel sums
// of eat_bn_act_bwd_reduce / eat_se_bn_bwd_combine.  Blocks without an expand conv: in_a = 1, in_b = 0, in_act = none,
// gpart may be NULL.  Only the geometries of the merged kernel:
// eat_dw_bwd_merged_ok(...) != 0, else EAT_EINVAL.
static int dw_bwd_bn_geometry_ok(int B, int C, int F, int T, int Fo, int To, int k, int stride) {
  if ((long long)B * C > 0x3fffffffLL || (long long)F * T >= (1 << 28)) return 0;
  if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return 0;
  if (stride == 1 && (Fo != F || To != T)) return 0;
  if ((long long)4 * C * F * T * 4 >= 0x7fffffffLL) return 0;         // lane-group offsets inside a wave's samples are 32-bit
  return 1;
}

// Host helper: 1 where eat_dw_conv_bwd_bn_g runs AND is the faster plan (EAT_DW_BN_K5=0: 5x5 convs keep the apply pass +
// eat_dw_conv_bwd_g - their on-load instances sit at 240 registers and gain 4 % only).
extern "C" int eat_dw_bwd_merged_ok(int B, int C, int F, int T, int Fo, int To, int k, int stride) {
  return dw_bwd_bn_geometry_ok(B, C, F, T, Fo, To, k, stride);
}

extern "C" int eat_dw_conv_bwd_bn_g(const float* dy, const float* z, const float* bn_a, const float* bn_b,
                                    const float* bn_mean, const float* bn_invstd, const float* gscale, const float* gadd,
                                    const double* sums, int bn_act, int frozen, const float* x, const float* in_a,
                                    const float* in_b, int in_act, const float* w, float* g, float* dw,
                                    float* gpart, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo, int To,
                                    int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dy || !z || !bn_a || !bn_b || !bn_mean || !bn_invstd || !sums || !in_a || !in_b)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g: missing operand");
  if (in_act < 0 || in_act > 2 || bn_act < 0 || bn_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g: bad act");
  if (!dw_bwd_bn_geometry_ok(B, C, F, T, Fo, To, k, stride))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g: geometry not covered by the merged kernel (F=%d T=%d k=%d stride=%d)", F, T, k, stride);
  if (gpart && inner_cap < eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g: partial buffer too small (inner_cap %d)", inner_cap);
  const eat::DwBnBwd bn{z, bn_a, bn_b, bn_mean, bn_invstd, gscale, gadd, sums, bn_act, frozen};
  const int rc = eat::dw_bwd_try(dy, x, in_a, in_b, in_act, w, g, dw, gpart, h_inner, B, C, F, T, Fo, To, k, stride,
                                 (hipStream_t)stream, &bn);
  if (rc == 1) return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g: merged kernel unavailable");
  return rc;
}

// The same over bf16-stored dy, z, x and g (act_io.h; the bf16-storage plan of BASELINE configs[2]): every wide tensor the
// backward of a block touches is 16-bit in HBM (x_b16 = 0: x and g are fp32 - the first block, whose conv input is the stem
// output); coefficients, channel sums, taps, dw and the partial sums (taken of the g values as stored) are fp32 / fp64.
extern "C" int eat_dw_conv_bwd_bn_g_b16(const void* dy, const void* z, const float* bn_a, const float* bn_b,
                                        const float* bn_mean, const float* bn_invstd, const float* gscale, const float* gadd,
                                        const double* sums, int bn_act, int frozen, const void* x, int x_b16, const float* in_a,
                                        const float* in_b, int in_act, const float* w, void* g, float* dw, float* gpart,
                                        int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo, int To, int k,
                                        int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dy || !z || !bn_a || !bn_b || !bn_mean || !bn_invstd || !sums || !x || !in_a || !in_b || !w || !g || !dw)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g_b16: missing operand");
  if (in_act < 0 || in_act > 2 || bn_act < 0 || bn_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g_b16: bad act");
  if (!dw_bwd_bn_geometry_ok(B, C, F, T, Fo, To, k, stride) || (F * T) % 2 != 0 || (Fo * To) % 2 != 0)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g_b16: geometry not covered by the merged kernel (F=%d T=%d k=%d stride=%d)", F, T, k, stride);
  if (gpart && inner_cap < eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g_b16: partial buffer too small (inner_cap %d)", inner_cap);
  const eat::DwBnBwd bn{reinterpret_cast<const float*>(z), bn_a, bn_b, bn_mean, bn_invstd, gscale, gadd, sums, bn_act, frozen};
  const int rc = eat::dw_bwd_try(reinterpret_cast<const float*>(dy), reinterpret_cast<const float*>(x), in_a, in_b, in_act, w,
                                 reinterpret_cast<float*>(g), dw, gpart, h_inner, B, C, F, T, Fo, To, k, stride,
                                 (hipStream_t)stream, &bn, 0, nullptr, nullptr, x_b16 ? 1 : 2);
  if (rc == 1) return eat::fail(EAT_EINVAL, "eat_dw_conv_bwd_bn_g_b16: merged kernel unavailable");
  return rc;
}

// The same for DyMN's dynamic depthwise conv (per-(b,c) taps w_bc (B, C, k*k), models/dymn/dy_block.py:103-131 backward):
// dw_bc (B, C, k*k) receives the per-plane tap gradients (zero-filled by the caller: planes of several tiles are added),
// res (shape of g) or NULL is added to g after the partial sums are taken (the skip connection of a block without expand
// conv), gzpart (layout of gpart) or NULL receives the per-tile sums of g * x (x raw): with gpart the two sums the
// BatchNorm backward of the expand conv needs - no reduce pass over (g, x).
extern "C" int eat_dw_conv_dyn_bwd_bn_g(const float* dy, const float* z, const float* bn_a, const float* bn_b,
                                        const float* bn_mean, const float* bn_invstd, const double* sums, int bn_act,
                                        int frozen, const float* x, const float* in_a, const float* in_b, int in_act,
                                        const float* w_bc, const float* res, float* g, float* dw_bc, float* gpart,
                                        float* gzpart, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo,
                                        int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dy || !z || !bn_a || !bn_b || !bn_mean || !bn_invstd || !sums || !in_a || !in_b || !w_bc || !g || !dw_bc)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g: missing operand");
  if (in_act < 0 || in_act > 2 || bn_act < 0 || bn_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g: bad act");
  if (!dw_bwd_bn_geometry_ok(B, C, F, T, Fo, To, k, stride))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g: geometry not covered by the merged kernel (F=%d T=%d k=%d stride=%d)", F, T, k, stride);
  if ((gpart || gzpart) && inner_cap < eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g: partial buffer too small (inner_cap %d)", inner_cap);
  const eat::DwBnBwd bn{z, bn_a, bn_b, bn_mean, bn_invstd, nullptr, nullptr, sums, bn_act, frozen};
  const int rc = eat::dw_bwd_try(dy, x, in_a, in_b, in_act, w_bc, g, dw_bc, gpart, h_inner, B, C, F, T, Fo, To, k, stride,
                                 (hipStream_t)stream, &bn, 1, res, gzpart);
  if (rc == 1) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g: merged kernel unavailable");
  return rc;
}

// ... over bf16-stored dy, z (and x, g when x_b16 != 0; x_b16 = 0: the block without expand conv - x is the fp32 block input, g
// the fp32 input gradient, res its skip gradient): the DyMN blocks of the bf16-storage plan.  res needs x_b16 = 0.
extern "C" int eat_dw_conv_dyn_bwd_bn_g_b16(const void* dy, const void* z, const float* bn_a, const float* bn_b,
                                            const float* bn_mean, const float* bn_invstd, const double* sums, int bn_act,
                                            int frozen, const void* x, int x_b16, const float* in_a, const float* in_b, int in_act,
                                            const float* w_bc, const float* res, void* g, float* dw_bc, float* gpart,
                                            float* gzpart, int inner_cap, int* h_inner, int B, int C, int F, int T, int Fo,
                                            int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dy || !z || !bn_a || !bn_b || !bn_mean || !bn_invstd || !sums || !x || !in_a || !in_b || !w_bc || !g || !dw_bc)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g_b16: missing operand");
  if (in_act < 0 || in_act > 2 || bn_act < 0 || bn_act > 2) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g_b16: bad act");
  if (res && x_b16) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g_b16: the skip gradient goes with an fp32 g (x_b16 = 0)");
  if (!dw_bwd_bn_geometry_ok(B, C, F, T, Fo, To, k, stride) || (F * T) % 2 != 0 || (Fo * To) % 2 != 0)
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g_b16: geometry not covered by the merged kernel (F=%d T=%d k=%d stride=%d)", F, T, k, stride);
  if ((gpart || gzpart) && inner_cap < eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride))
    return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g_b16: partial buffer too small (inner_cap %d)", inner_cap);
  const eat::DwBnBwd bn{reinterpret_cast<const float*>(z), bn_a, bn_b, bn_mean, bn_invstd, nullptr, nullptr, sums, bn_act, frozen};
  const int rc = eat::dw_bwd_try(reinterpret_cast<const float*>(dy), reinterpret_cast<const float*>(x), in_a, in_b, in_act, w_bc,
                                 reinterpret_cast<float*>(g), dw_bc, gpart, h_inner, B, C, F, T, Fo, To, k, stride,
                                 (hipStream_t)stream, &bn, 1, res, gzpart, x_b16 ? 1 : 2);
  if (rc == 1) return eat::fail(EAT_EINVAL, "eat_dw_conv_dyn_bwd_bn_g_b16: no instance for F=%d T=%d k=%d stride=%d x_b16=%d", F, T, k, stride, x_b16);
  return rc;
}

extern "C" int eat_dw_conv_dyn_wgrad(const float* dz, const float* x, float* dw_bc, int B, int C, int F, int T, int Fo,
                                     int To, int k, int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  return dw_wgrad_impl(dz, x, dw_bc, B, C, C, F, T, Fo, To, k, stride, 1, stream);
}

// Launch plan of the 1x1 weight gradient: which kernel, how the k range is cut (also exported through
// eat_pw_wgrad_slots so that a caller can size a one-slot-per-block workspace)
struct WgPlan { int kind; int upb; unsigned nz; int sps; int bpb; int mtb, ntb, mg, ng; bool gram; int ps_spl;
                int w_ptr = 0, w_qtr = 0, w_ptn = 0, w_qtn = 0, w_depth = 0; bool w_swap = false; };
// kind: 0 LDS-free streaming kernel (thin matrices), 1 LDS-staged x3, 2 exact fp32, 3 wide-tile LDS ring (pw_wgrad_wide_kernel);
// mtb / ntb: row tiles per block, mg / ng groups; w_*: tile rows, tile counts, ring depth and operand order of kind 3

// (row tiles per block) pairs the streaming kernel is instantiated for
// (the x side carries the SE scale / BatchNorm transform and costs more registers per tile: <= 3 tiles there, <= 4 on the dz side)
static bool thin_pair(int m, int n) { return m >= 1 && m <= 4 && n >= 1 && n <= 3; }
// Tile shape of the wide-tile kernel for a (Co, Ci) matrix, and whether the plan uses it (EAT_WGRAD_WIDE: bit 0 = instead of
// the 128 x 128-tile kernel, bit 1 = also instead of the streaming kernel where that re-reads an operand; default 3).
// Measured (tools/bench_kernels.py wgrad): the producers' fixed cost per unit (13 load instructions, one barrier) loses on
// tiles of fewer than ~20 pieces (160 rows of P + Q), and the on-load transform makes the producers the pole.
struct WideShape { bool ok; bool swap; int ptr, qtr, ptn, qtn; };
static WideShape wide_shape(int Co, int Ci, bool per_sample, bool same, bool no_wide, bool has_xscale, bool has_tf) {
  constexpr int wide_min = 20;
  WideShape w{false, Ci > Co, 0, 0, 0, 0};
  if (per_sample || no_wide || has_tf || (has_xscale && (Ci & 3) != 0)) return w;
  if (same) {
    // Gram matrix (dz == x, train_fuse.hip): ONE operand, loaded once - P = x, the Q fragments are read from P's rows.  Above
    // the streaming kernel's range (C > 64) up to what one consumer quartet holds (10 column tiles); 80 x 80 at 504
    // positions x 256 clips: 62 us on the 128 x 128-tile kernel for 41 MB of input
    if (Co != Ci || Co <= 64 || Co > 160 || has_xscale) return w;
    w.swap = true;
    w.ptn = w.qtn = 1;
    w.ptr = w.qtr = (Co + 15) / 16 * 16;
    w.ok = true;
    return w;
  }
  const int PR = w.swap ? Ci : Co, QR = w.swap ? Co : Ci;
  // (tile limits measured: P <= 192 / 128 rows per block instead of 256: 672 x 112 147 -> 157 / 198 us, mn10 step +0.3 ms;
  //  384 blocks instead of one per CU: +0.3 ms; minimum of 14 / 30 pieces instead of 20: +0.1 ms)
  w.ptn = (PR + 255) / 256;
  w.ptr = ((PR + w.ptn - 1) / w.ptn + 15) / 16 * 16;
  w.qtn = (QR + 159) / 160;                                  // rows of Q per block: 4 producers x 5 pieces of 8 rows
  w.qtr = ((QR + w.qtn - 1) / w.qtn + 15) / 16 * 16;
  w.ok = (w.ptn - 1) * w.ptr < PR && (w.qtn - 1) * w.qtr < QR && w.ptr / 8 + w.qtr / 8 >= wide_min;
  return w;
}
static WgPlan wgrad_plan(int B, int Co, int Ci, int S, int per_sample, int exact_fp32, bool same, bool has_scale_or_tf,
                         bool has_xscale = false, bool no_wide = false,
                         bool has_tf = false) {
  static const bool env_fp32 = getenv("EAT_WGRAD_FP32") && atoi(getenv("EAT_WGRAD_FP32")) != 0;
  const bool force_fp32 = env_fp32 || exact_fp32 == 1;      // exact_fp32: 0 = bf16x3, 1 = exact fp32, 2 = plain bf16
  const bool ps_x3 = per_sample && Co >= 64 && Ci >= 64;
  WgPlan p{2, 0, 0, (S + 31) / 32, 0, 0, 0, 1, 1, false, 0};
  if (!force_fp32 && (S & 3) == 0) {
    const int sps = p.sps;
    const int mtn = (Co + 15) / 16, ntn = (Ci + 15) / 16;
    if (per_sample && !ps_x3 && sps >= 32) {
      // per-sample gradients of the thin early-layer matrices (one side < 64 channels, planes of >= 1024 positions):
      // the same streaming kernel, a few blocks per sample adding into the sample's own matrix
      const int mg = (mtn + 3) / 4, ng = (ntn + 2) / 3;
      const int mtb = (mtn + mg - 1) / mg, ntb = (ntn + ng - 1) / ng;
      if (mg * ng <= 4 && thin_pair(mtb, ntb)) {
        int spl = 1024 / (mg * ng * B);
        if (spl > sps / 16) spl = sps / 16;
        if (spl < 1) spl = 1;
        p.kind = 0; p.mtb = mtb; p.ntb = ntb; p.mg = mg; p.ng = ng;
        p.upb = (sps + spl - 1) / spl;
        p.ps_spl = (sps + p.upb - 1) / p.upb;
        p.nz = (unsigned)(B * p.ps_spl);
        return p;
      }
    }
  }
  if (!force_fp32 && (!per_sample || ps_x3) && (S & 3) == 0) {
    const int sps = p.sps;
    const int tiles = ((Co + 127) / 128) * ((Ci + 127) / 128);
    const long long total = (long long)B * sps;
    const bool gram = same && Co == Ci && !has_scale_or_tf;
    const int mtn = (Co + 15) / 16, ntn = (Ci + 15) / 16;
    bool thin = false;
    if (!per_sample) {
      if (gram && Co <= 64) {                                  // Gram matrix: the operand is loaded once
        thin = true; p.mtb = p.ntb = mtn; p.mg = p.ng = 1; p.gram = true;
      } else if (Co <= 64 && Ci <= 64 && (Co <= 16 || Ci <= 16)) {
        thin = true; p.mtb = mtn; p.ntb = ntn; p.mg = p.ng = 1;
      } else if (total * 32 >= (1 << 19)) {
        // a long k axis (>= 512 k positions) over few rows: groups of <= 4 row tiles per block, at most 4 groups (the other
        // operand is re-read once per group, from L2)
        const int mg = (mtn + 3) / 4, ng = (ntn + 2) / 3;
        const int mtb = (mtn + mg - 1) / mg, ntb = (ntn + ng - 1) / ng;
        if (mg * ng <= 4 && thin_pair(mtb, ntb)) { thin = true; p.mtb = mtb; p.ntb = ntb; p.mg = mg; p.ng = ng; }
        // more than one row group = the other operand is read once per group: the wide-tile kernel reads it once
        if (thin && mg * ng > 1 && wide_shape(Co, Ci, per_sample, same, no_wide, has_xscale, has_tf).ok) thin = false;
      }
    }
    if (thin) {
      long long splits = (1024 / (p.mg * p.ng)) < total ? (1024 / (p.mg * p.ng)) : total;
      p.kind = 0;
      p.upb = (int)((total + splits - 1) / splits);
      p.nz = (unsigned)((total + p.upb - 1) / p.upb);
      return p;
    }
    {
      const WideShape w = wide_shape(Co, Ci, per_sample, same, no_wide, has_xscale, has_tf);
      if (w.ok) {
        p.w_swap = w.swap; p.w_ptr = w.ptr; p.w_qtr = w.qtr; p.w_ptn = w.ptn; p.w_qtn = w.qtn;
        const int wtiles = w.ptn * w.qtn;
        constexpr int wtarget = 256;                           // one block per CU
        long long splits = wtiles >= wtarget ? 1 : wtarget / wtiles;
        if (splits > total / 16) splits = total / 16;
        if (splits < 1) splits = 1;
        p.kind = 3;
        p.upb = (int)((total + splits - 1) / splits);
        p.nz = (unsigned)((total + p.upb - 1) / p.upb);
        return p;
      }
    }
    p.kind = 1;
    p.upb = sps;                                             // per-sample gradients: one sample per block
    if (!per_sample) {
      // ~512 blocks, but at least 16 units (512 k) of MFMA work in front of a block's Co x Ci atomics
      // (512 = one round of two resident blocks per CU; 1024 measured 0.26 ms slower per mn10 step: the second round pays
      // prologue, tail and the Co x Ci atomics again)
      constexpr int target = 512;
      // never MORE than `target` blocks: 516 blocks (6 tiles x 86 slices, the 672 x 112 layers) ran as a full round of 512
      // resident blocks plus a second round of 4 (183 -> 158 us with 510)
      long long splits = tiles >= target ? 1 : target / tiles;
      if (splits > total / 16) splits = total / 16;
      if (splits < 1) splits = 1;
      p.upb = (int)((total + splits - 1) / splits);
    }
    p.nz = (unsigned)((total + p.upb - 1) / p.upb);
    return p;
  }
  const int tiles = ((Co + 31) / 32) * ((Ci + 31) / 32);
  int splits = (1024 + tiles - 1) / tiles;
  if (splits > B || per_sample) splits = B;
  p.bpb = (B + splits - 1) / splits;
  p.nz = (unsigned)((B + p.bpb - 1) / p.bpb);
  return p;
}

static int pw_wgrad_impl(const float* dz, const float* x, const float* x_scale, float* dW, int B, int Co, int Ci, int S,
                         int per_sample, int exact_fp32, eat_stream_t stream, float* ws = nullptr, int n_slots = 0,
                         WgTf tf = WgTf{nullptr, nullptr, 0}) {
  // default: split-operand bf16 MFMA kernel (fp32-class accuracy); exact_fp32 (the caller's precision choice),
  // EAT_WGRAD_FP32=1 (process-wide debug override) or S % 4 != 0: exact fp32 MFMA kernel.
  // per-sample gradients (DyMN: K = one plane, B x Co x Ci outputs): the bf16x3 kernel with one block per (tile, sample)
  // and plain stores from Co, Ci >= 64 on (EAT_DYN_WGRAD_X3=0 restores the 32 x 32-tile fp32 kernel).
  // ws / n_slots (zero-filled, n_slots * Co * Ci floats): the blocks' atomics go to copy blockIdx.z % n_slots and a
  // second kernel adds the copies into dW in a fixed order; n_slots >= eat_pw_wgrad_slots(...) gives every block its own
  // copy (bit-reproducible result).  The LDS-staged and exact kernels use the workspace only in that one-per-block form.
  // (a centring transform - tf.actr set - keeps the Gram plan: both operands are the same centred rows)
  // centring form of the Gram launches (eat_gram_centered): a = 1, b = actr = -mean, no activation - an additive row constant
  const bool centring = dz == x && tf.actr != nullptr && tf.act == EAT_ACT_NONE;
  WgPlan p = wgrad_plan(B, Co, Ci, S, per_sample, exact_fp32, dz == x, x_scale != nullptr || (tf.a != nullptr && !tf.actr),
                        x_scale != nullptr, tf.actr != nullptr && !centring, tf.a != nullptr && !centring);
  // the wide-tile kernel stores one copy of dW per k-slice: it needs the workspace (eat_pw_conv_wgrad_ws with
  // n_slots >= eat_pw_wgrad_slots) and 16-byte aligned rows; without them the plan is the one without it
  // A caller that DID bring a workspace sized it with eat_pw_wgrad_slots / eat_pw_wgrad_kernel_kind and - for this kernel -
  // left it uninitialised: too few copies means the two plans diverged, and the atomic kernels of the fallback plan would
  // add into garbage.  Fail instead of falling back.
  if (p.kind == 3 && ws != nullptr && (Ci & 3) == 0 && n_slots < (int)p.nz)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_ws: workspace of %d copies, the stored-slice kernel needs %u "
                     "(eat_pw_wgrad_slots)", n_slots, p.nz);
  if (p.kind == 3 && !(ws != nullptr && n_slots >= (int)p.nz && (Ci & 3) == 0))
    p = wgrad_plan(B, Co, Ci, S, per_sample, exact_fp32, dz == x, x_scale != nullptr || (tf.a != nullptr && !tf.actr),
                   x_scale != nullptr, true, tf.a != nullptr && !centring);
  hipStream_t hs = (hipStream_t)stream;
  const bool priv = ws != nullptr && !per_sample && n_slots >= (int)p.nz;     // one copy per block
  if (p.kind == 0) {
    const bool use_ws = ws != nullptr && n_slots > 1;
    float* target = use_ws ? ws : dW;
    const int slots = use_ws ? (n_slots < (int)p.nz ? n_slots : (int)p.nz) : 1;
#define EAT_NARROW(M_, N_) if (p.mtb == M_ && p.ntb == N_) launch_narrow<M_, N_>(dz, x, x_scale, target, B, Co, Ci, S, p.sps, p.upb, p.nz, slots, hs, tf, p.mg, p.ng, p.gram, p.ps_spl)
    EAT_NARROW(1, 1); EAT_NARROW(1, 2); EAT_NARROW(1, 3); EAT_NARROW(1, 4);
    EAT_NARROW(2, 1); EAT_NARROW(3, 1); EAT_NARROW(4, 1);
    EAT_NARROW(2, 2); EAT_NARROW(3, 3); EAT_NARROW(4, 4);       // (2,2), (4,4): Gram mode only (dz == x, Co == Ci)
    EAT_NARROW(2, 3); EAT_NARROW(3, 2); EAT_NARROW(4, 2); EAT_NARROW(4, 3);
#undef EAT_NARROW
    if (use_ws)
      hipLaunchKernelGGL(wgrad_slot_reduce_kernel, dim3((Co * Ci + 63) / 64), dim3(256), 0, hs, ws, dW, Co * Ci, slots);
    return eat::check_launch("eat_pw_conv_wgrad");
  }
  float* target = priv ? ws : dW;
  const int slots = priv ? (int)p.nz : 0;
  if (p.kind == 3) {
    const size_t smem = (size_t)(p.w_ptr / 8 + p.w_qtr / 8) * 2 * 1024;     // two slots of converted fragments
    dim3 grid(p.w_ptn, p.w_qtn, p.nz);
    // (the phase-decomposition switches of the round-4 measurement builds - no epilogue stores, no MFMAs, no loads, no
    //  conversion: DESIGN 3.15 - are not in the shipped kernel)
#define EAT_WIDE(NP_, SW_, SC_, RA_)                                                                                      \
    do {                                                                                                                  \
      auto kern = pw_wgrad_wide_kernel<NP_, SW_, SC_, RA_>;                                                               \
      static bool attr_set = false;                                                                                       \
      if (!attr_set) {                                                                                                    \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
          return eat::fail(EAT_ELAUNCH, "eat_pw_conv_wgrad: hipFuncSetAttribute(160 KB of LDS) failed");                  \
        attr_set = true;                                                                                                  \
      }                                                                                                                   \
      hipLaunchKernelGGL(kern, grid, dim3(512), smem, hs, dz, x, x_scale, target, B, Co, Ci, S, p.sps, p.upb, p.w_ptr,    \
                         p.w_qtr, centring ? tf.b : (const float*)nullptr, dz == x ? 1 : 0, (const float*)nullptr, \
                         (const float*)nullptr, 0, 0);                                                                   \
    } while (0)
#define EAT_WIDE_SC(NP_, SW_) do { if (x_scale) EAT_WIDE(NP_, SW_, true, false); else EAT_WIDE(NP_, SW_, false, false); } while (0)
#define EAT_WIDE_SW(NP_)                                                                                                  \
    do {                                                                                                                  \
      if (centring) EAT_WIDE(NP_, true, false, true);                /* Gram matrices: P = x, no SE scale */                \
      else if (p.w_swap) EAT_WIDE_SC(NP_, true);                                                                          \
      else EAT_WIDE_SC(NP_, false);                                                                                       \
    } while (0)
    if (exact_fp32 == 2) EAT_WIDE_SW(1); else EAT_WIDE_SW(3);
#undef EAT_WIDE_SC
#undef EAT_WIDE_SW
#undef EAT_WIDE
  } else if (p.kind == 1) {
    dim3 grid((Co + 127) / 128, (Ci + 127) / 128, p.nz);
    if (exact_fp32 == 2)
      hipLaunchKernelGGL(pw_wgrad_x3_kernel<1>, grid, dim3(256), 0, hs, dz, x, x_scale, target, B, Co, Ci, S, p.sps, p.upb,
                         per_sample, tf, slots);
    else
      hipLaunchKernelGGL(pw_wgrad_x3_kernel<3>, grid, dim3(256), 0, hs, dz, x, x_scale, target, B, Co, Ci, S, p.sps, p.upb,
                         per_sample, tf, slots);
  } else {
    dim3 grid((Co + 31) / 32, (Ci + 31) / 32, p.nz);
    hipLaunchKernelGGL(pw_wgrad_kernel, grid, dim3(256), 0, hs, dz, x, x_scale, target, B, Co, Ci, S, p.bpb, per_sample, tf,
                       slots);
  }
  if (priv && p.kind == 3)
    hipLaunchKernelGGL(wgrad_slot_reduce4_kernel, dim3((Co * Ci + 255) / 256), dim3(256), 0, hs, ws, dW, Co * Ci, slots);
  else if (priv)
    hipLaunchKernelGGL(wgrad_slot_reduce_kernel, dim3((Co * Ci + 63) / 64), dim3(256), 0, hs, ws, dW, Co * Ci, slots);
  return eat::check_launch("eat_pw_conv_wgrad");
}

// Number of workspace copies that gives every block of eat_pw_conv_wgrad_ws its own (bit-reproducible result); `same`: the
// two operands are the same tensor (Gram matrix).  Host helper.
static int eat_pw_wgrad_kernel_kind_nowide(int B, int Co, int Ci, int S, int exact_fp32, int same, int has_scale, int has_tf) {
  const WgPlan p = wgrad_plan(B, Co, Ci, S, 0, exact_fp32, same != 0, has_scale != 0 || has_tf != 0, has_scale != 0, true,
                              has_tf != 0);
  return p.kind == 0 ? 10 * (1000 * p.mtb + 10 * p.ntb + (p.gram ? 1 : 0)) : p.kind;
}
// Which kernel family eat_pw_conv_wgrad[_ws|_tf] launches for a shape (host helper for bench.py's byte models and the
// profiles): 0 = pw_wgrad_x3_narrow_kernel<mtb, ntb> (returned as 1000 * mtb + 10 * ntb + gram), 1 = pw_wgrad_x3_kernel,
// 2 = pw_wgrad_kernel (exact fp32), 3 = pw_wgrad_wide_kernel; encoded as kind + 10 * detail.
extern "C" int eat_pw_wgrad_kernel_kind(int B, int Co, int Ci, int S, int exact_fp32, int same, int has_scale, int has_tf) {
  const WgPlan p = wgrad_plan(B, Co, Ci, S, 0, exact_fp32, same != 0, has_scale != 0 || has_tf != 0, has_scale != 0, false,
                              has_tf != 0);
  if (p.kind == 3 && (Ci & 3) != 0)                           // (the wide-tile kernel needs 16-byte aligned rows of dW)
    return eat_pw_wgrad_kernel_kind_nowide(B, Co, Ci, S, exact_fp32, same, has_scale, has_tf);
  if (p.kind == 0) return 10 * (1000 * p.mtb + 10 * p.ntb + (p.gram ? 1 : 0));
  return p.kind;
}

extern "C" int eat_pw_wgrad_slots(int B, int Co, int Ci, int S, int exact_fp32, int same) {
  return (int)wgrad_plan(B, Co, Ci, S, 0, exact_fp32, same != 0, false).nz;
}

extern "C" int eat_pw_conv_wgrad(const float* dz, const float* x, const float* x_scale, float* dW, int B, int Co,
                                 int Ci, int S, int exact_fp32, eat_stream_t stream) {
  eat::clear_stale_error();
  return pw_wgrad_impl(dz, x, x_scale, dW, B, Co, Ci, S, 0, exact_fp32, stream);
}

// The same with a zero-filled workspace ws of n_slots * Co * Ci floats: the streaming kernel of the small matrices spreads
// its atomics over the slots (same-address atomics serialise in L2) and a second kernel adds the slots into dW.
// Matrices that do not run on that kernel ignore ws.
extern "C" int eat_pw_conv_wgrad_ws(const float* dz, const float* x, const float* x_scale, float* dW, float* ws,
                                    int n_slots, int B, int Co, int Ci, int S, int exact_fp32, eat_stream_t stream) {
  eat::clear_stale_error();
  return pw_wgrad_impl(dz, x, x_scale, dW, B, Co, Ci, S, 0, exact_fp32, stream, ws, n_slots);
}

// ... and with the x operand act(tf_a[ci] * x + tf_b[ci]) evaluated on load (x_scale multiplies the transformed value):
// the weight gradient of a project conv that was run by eat_pw_conv_tf_fwd.  ws / n_slots as above (may be NULL / 0).
extern "C" int eat_pw_conv_wgrad_tf(const float* dz, const float* x, const float* tf_a, const float* tf_b, int tf_act,
                                    const float* x_scale, float* dW, float* ws, int n_slots, int B, int Co, int Ci, int S,
                                    int exact_fp32, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!tf_a || !tf_b) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_tf: tf_a and tf_b are required");
  if (tf_act < 0 || tf_act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_tf: bad act %d", tf_act);
  return pw_wgrad_impl(dz, x, x_scale, dW, B, Co, Ci, S, 0, exact_fp32, stream, ws, n_slots, WgTf{tf_a, tf_b, tf_act});
}

// ---- 1x1 weight gradient of the bf16-storage plan (act_io.h; BASELINE configs[2]): ONE of the operands - the wide tensor -
// is bf16 in HBM: x (the project conv's input y_d, or z_d with act(tf_a x + tf_b) evaluated on load, times x_scale) or dz (the
// expand conv's incoming gradient g).  Plain bf16 products, fp32 accumulation (what autocast does to the conv weight
// gradient, ex_pl_audioset.py:287-293).  Always the wide-tile producer / consumer kernel with the bf16 operand as P (its
// fragments need no conversion); ws: >= eat_pw_wgrad_b16_slots(...) * Co * Ci floats (no zero fill), dW is added to (zeroed by
// the caller).  S % 4 == 0, Ci % 4 == 0.
struct WgB16Plan { WideShape w; int upb; unsigned nz; int sps; };
static WgB16Plan wgrad_b16_plan(int B, int Co, int Ci, int S, int x_b16) {
  // x_b16: 1 = x is the bf16 (wide) operand, 0 = dz is; 2 = BOTH operands fp32 (the per-sample gradients of the fp32-storage
  // DyMN plan on the same kernel, split-operand products): P = the operand with more rows
  WgB16Plan p{};
  p.sps = (S + 31) / 32;
  WideShape& w = p.w;
  w.swap = x_b16 == 2 ? Ci > Co : x_b16 != 0;
  const int PR = w.swap ? Ci : Co, QR = w.swap ? Co : Ci;
  w.ptn = (PR + 255) / 256;
  w.ptr = ((PR + w.ptn - 1) / w.ptn + 15) / 16 * 16;
  w.qtn = (QR + 159) / 160;
  w.qtr = ((QR + w.qtn - 1) / w.qtn + 15) / 16 * 16;
  w.ok = (w.ptn - 1) * w.ptr < PR && (w.qtn - 1) * w.qtr < QR;
  const long long total = (long long)B * p.sps;
  const int wtiles = w.ptn * w.qtn;
  long long splits = wtiles >= 256 ? 1 : 256 / wtiles;                // one block per CU
  if (splits > total / 16) splits = total / 16;
  if (splits < 1) splits = 1;
  p.upb = (int)((total + splits - 1) / splits);
  p.nz = (unsigned)((total + p.upb - 1) / p.upb);
  return p;
}

extern "C" int eat_pw_wgrad_b16_slots(int B, int Co, int Ci, int S, int x_b16) {
  return (int)wgrad_b16_plan(B, Co, Ci, S, x_b16).nz;
}

extern "C" int eat_pw_conv_wgrad_b16(const void* dz, int dz_b16, const void* x, int x_b16, const float* tf_a,
                                     const float* tf_b, int tf_act, const float* x_scale, float* dW, float* ws, int n_slots,
                                     int B, int Co, int Ci, int S, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dz || !x || !dW || !ws) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: missing operand");
  if ((dz_b16 != 0) == (x_b16 != 0)) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: exactly one of dz / x is the bf16 (wide) tensor");
  if (B < 1 || Co < 1 || Ci < 4 || (Ci & 3) != 0 || S < 4 || (S & 3) != 0)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: Ci=%d and S=%d must be multiples of 4", Ci, S);
  if ((tf_a == nullptr) != (tf_b == nullptr) || tf_act < 0 || tf_act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: bad transform");
  if (dz_b16 && (tf_a || x_scale)) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: transform / scale belong to a bf16 x operand");
  const WgB16Plan p = wgrad_b16_plan(B, Co, Ci, S, x_b16);
  if (!p.w.ok) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: internal tiling error (%d x %d)", Co, Ci);
  if (n_slots < (int)p.nz) return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: workspace of %d copies, %u needed", n_slots, p.nz);
  if ((long long)(x_b16 ? Ci : Co) * S * 2 > 0x7fffffffLL || (long long)(x_b16 ? Co : Ci) * S * 4 > 0x7fffffffLL)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_wgrad_b16: a sample exceeds the 32-bit row offsets");
  hipStream_t hs = (hipStream_t)stream;
  const size_t smem = (size_t)(p.w.ptr / 8 + p.w.qtr / 8) * 2 * 1024;
  dim3 grid(p.w.ptn, p.w.qtn, p.nz);
  const float* fdz = reinterpret_cast<const float*>(dz);
  const float* fx = reinterpret_cast<const float*>(x);
#define EAT_WIDE16(SW_, SC_, TF_)                                                                                          \
  do {                                                                                                                    \
    auto kern = pw_wgrad_wide_kernel<1, SW_, SC_, false, true, TF_>;                                                      \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
        return eat::fail(EAT_ELAUNCH, "eat_pw_conv_wgrad_b16: hipFuncSetAttribute(160 KB of LDS) failed");                \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, hs, fdz, fx, x_scale, ws, B, Co, Ci, S, p.sps, p.upb, p.w.ptr, p.w.qtr,    \
                       (const float*)nullptr, 0, tf_a, tf_b, tf_act, 0);                                                  \
  } while (0)
  if (!x_b16) EAT_WIDE16(false, false, false);
  else if (tf_a && x_scale) EAT_WIDE16(true, true, true);
  else if (tf_a) EAT_WIDE16(true, false, true);
  else if (x_scale) EAT_WIDE16(true, true, false);
  else EAT_WIDE16(true, false, false);
#undef EAT_WIDE16
  hipLaunchKernelGGL(wgrad_slot_reduce4_kernel, dim3((Co * Ci + 255) / 256), dim3(256), 0, hs, ws, dW, Co * Ci, (int)p.nz);
  return eat::check_launch("eat_pw_conv_wgrad_b16");
}

// Per-sample weight gradients of a dynamic 1x1 conv under the bf16-storage plan (autograd of the grouped F.conv2d of
// models/dymn/dy_block.py:120-127): dW_b (B, Co, Ci) = dz[b] x[b]^T with exactly one bf16 operand (the wide tensor), plain bf16
// products, fp32 accumulation.  The wide-tile kernel of eat_pw_conv_wgrad_b16 with one k-slice per SAMPLE: slice b is stored as
// dW_b[b] - every element of dW_b is written, no zero fill, no reduction.  S % 4 == 0, Ci % 4 == 0.
// k-slices per sample: a sample's reduction is cut into several blocks where B x (tiles of dW) alone would leave CUs idle - the
// early layers (thin matrices, planes of thousands of positions: 128 one-tile blocks walking 1000 units each ran at 1.9 TB/s)
static int dyn_wgrad_b16_slices(const WgB16Plan& p, int B) {
  const long long blocks = (long long)p.w.ptn * p.w.qtn * B;
  int ns = 1;
  while (ns < 8 && blocks * ns < 1024 && p.sps / (2 * ns) >= 8) ns *= 2;
  while (ns > 1 && (ns - 1) * ((p.sps + ns - 1) / ns) >= p.sps) --ns;    // every slice must own at least one unit
  return ns;
}

extern "C" int eat_pw_dyn_wgrad_b16_slices(int B, int Co, int Ci, int S, int x_b16) {
  const WgB16Plan p = wgrad_b16_plan(B, Co, Ci, S, x_b16);
  if (!p.w.ok || (S & 3) != 0 || (Ci & 3) != 0) return 0;               // (0: this shape does not run on the wide-tile kernel)
  return dyn_wgrad_b16_slices(p, B);
}

extern "C" int eat_pw_conv_dyn_wgrad_b16(const void* dz, int dz_b16, const void* x, int x_b16, float* dW_b, int n_slices, int B,
                                         int Co, int Ci, int S, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dz || !x || !dW_b) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_wgrad_b16: missing operand");
  if (dz_b16 && x_b16) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_wgrad_b16: at most one of dz / x is a bf16 tensor");
  if (B < 1 || Co < 1 || Ci < 4 || (Ci & 3) != 0 || S < 4 || (S & 3) != 0)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_wgrad_b16: Ci=%d and S=%d must be multiples of 4", Ci, S);
  const bool f32 = !dz_b16 && !x_b16;                                  // both fp32: split-operand (bf16x3) products
  WgB16Plan p = wgrad_b16_plan(B, Co, Ci, S, f32 ? 2 : x_b16);
  if (!p.w.ok) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_wgrad_b16: internal tiling error (%d x %d)", Co, Ci);
  if ((long long)(Ci > Co ? Ci : Co) * S * 4 > 0x7fffffffLL)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_wgrad_b16: a sample exceeds the 32-bit row offsets");
  const int ns = dyn_wgrad_b16_slices(p, B);
  if (n_slices < ns) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_wgrad_b16: dW_b holds %d copies, %d needed (eat_pw_dyn_wgrad_b16_slices)", n_slices, ns);
  p.upb = (p.sps + ns - 1) / ns;                                       // ns k-slices per sample
  p.nz = (unsigned)(B * ns);
  hipStream_t hs = (hipStream_t)stream;
  const size_t smem = (size_t)(p.w.ptr / 8 + p.w.qtr / 8) * 2 * 1024;
  dim3 grid(p.w.ptn, p.w.qtn, p.nz);
  const float* fdz = reinterpret_cast<const float*>(dz);
  const float* fx = reinterpret_cast<const float*>(x);
#define EAT_WIDE16D(SW_)                                                                                                  \
  do {                                                                                                                    \
    auto kern = pw_wgrad_wide_kernel<1, SW_, false, false, true, false>;                                                  \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
        return eat::fail(EAT_ELAUNCH, "eat_pw_conv_dyn_wgrad_b16: hipFuncSetAttribute(160 KB of LDS) failed");            \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, hs, fdz, fx, (const float*)nullptr, dW_b, B, Co, Ci, S, p.sps, p.upb,  \
                       p.w.ptr, p.w.qtr, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, 0, ns);  \
  } while (0)
#define EAT_WIDE32D(SW_)                                                                                                  \
  do {                                                                                                                    \
    auto kern = pw_wgrad_wide_kernel<3, SW_, false, false>;                                                               \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
        return eat::fail(EAT_ELAUNCH, "eat_pw_conv_dyn_wgrad_b16: hipFuncSetAttribute(160 KB of LDS) failed");            \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, hs, fdz, fx, (const float*)nullptr, dW_b, B, Co, Ci, S, p.sps, p.upb,  \
                       p.w.ptr, p.w.qtr, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, 0, ns);  \
  } while (0)
  if (f32) { if (p.w.swap) EAT_WIDE32D(true); else EAT_WIDE32D(false); }
  else if (!x_b16) EAT_WIDE16D(false); else EAT_WIDE16D(true);
#undef EAT_WIDE16D
#undef EAT_WIDE32D
  if (ns > 1) {                                                        // copy 0 (B, Co, Ci) += copies 1 .. ns - 1, fixed order
    const long long n = (long long)B * Co * Ci;
    if (n > 0x7fffffffLL) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_wgrad_b16: B*Co*Ci exceeds the 32-bit index of the slice reduction");
    hipLaunchKernelGGL(wgrad_slot_reduce4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, dW_b + n, dW_b, (int)n, ns - 1);
  }
  return eat::check_launch("eat_pw_conv_dyn_wgrad_b16");
}

// 1 where eat_pw_conv_dyn_wgrad adds into dW_b (the caller zero-fills it), 0 where it stores.  Host helper.
extern "C" int eat_pw_dyn_wgrad_accumulates(int Co, int Ci, int S) {
  return wgrad_plan(1, Co, Ci, S, 1, 0, false, false).kind == 1 ? 0 : 1;
}

// Centred Gram matrix Gc = sum (x - m)(x - m)^T, m = sx * inv_n (see include/eat_hip.h): both operands are centred on
// load (operand transform a = 1, b = -m on the x side, additive row constant -m on the dz side; elements beyond the k
// range stay zero).
namespace {
__global__ void gram_center_coef_kernel(const float* __restrict__ sx, float inv_n, float* __restrict__ ab, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { ab[c] = 1.0f; ab[C + c] = -sx[c] * inv_n; }
}
}  // namespace
extern "C" int eat_gram_centered(const float* x, const float* sx, float inv_n, float* G, float* ws, int n_slots, int B, int C,
                                 int S, int exact_fp32, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !sx || !G || !ws || n_slots < 1) return eat::fail(EAT_EINVAL, "eat_gram_centered: missing operand (ws holds the (1, -m) coefficients first)");
  // ws: [2 C floats of transform coefficients][n_slots copies of G]
  float* ab = ws;
  hipLaunchKernelGGL(gram_center_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sx, inv_n, ab, C);
  return pw_wgrad_impl(x, x, nullptr, G, B, C, C, S, 0, exact_fp32, stream, ws + 2 * (size_t)C, n_slots,
                       WgTf{ab, ab + C, EAT_ACT_NONE, ab + C});
}

extern "C" int eat_pw_conv_dyn_wgrad(const float* dz, const float* x, float* dW_b, int B, int Co, int Ci, int S,
                                     eat_stream_t stream) {
  eat::clear_stale_error();
  return pw_wgrad_impl(dz, x, nullptr, dW_b, B, Co, Ci, S, 1, 0, stream);
}
