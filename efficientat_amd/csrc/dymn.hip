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
#include <cstdlib>
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
// Block = (16-row m-tile, sample).  The 16 rows of a bank are one contiguous run of 16*Ci floats, so they are read
// with coalesced loads in 256-column pieces, mixed with the sample's attention weights and parked in LDS (row pitch
// 257: the transposing reads below then hit 16 different banks); the MFMA A-fragment order is written back as
// 256-byte runs.  (Computing each packed element straight from global memory made every lane of a load touch a
// different row, i.e. a different cache line, for each of the K banks: 2.6 TB/s of packed output at best.)
constexpr int kPackCols = 256;
__global__ __launch_bounds__(256) void dyn_pw_pack_kernel(const float* __restrict__ bank, const float* __restrict__ att,
                                                          const float* __restrict__ row_scale, float* __restrict__ wp,
                                                          int K, int Co, int Ci, int MT) {
  __shared__ float s_w[16 * (kPackCols + 1)];
  const int mt = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int m0 = mt * 16;
  const int rows = (Co - m0) < 16 ? (Co - m0) : 16;
  const float* a = att + (size_t)b * K;
  const size_t N = (size_t)Co * Ci;
  float* out = wp + (size_t)b * (Ci / 4) * MT * 64;
  const int lane = tid & 63, wv = tid >> 6;
  const int m = lane & 15;
  const float rs = (row_scale && m0 + m < Co) ? row_scale[m0 + m] : 1.0f;
  for (int c0 = 0; c0 < Ci; c0 += kPackCols) {
    const int cols = (Ci - c0) < kPackCols ? (Ci - c0) : kPackCols;     // multiple of 4
    // phase 1: 16 x cols aggregated weights -> LDS (thread = (row, 4 columns))
    for (int e = tid; e < 16 * (cols >> 2); e += 256) {
      const int r = e / (cols >> 2), q = e - r * (cols >> 2);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows) {
        const size_t n = (size_t)(m0 + r) * Ci + c0 + 4 * q;
        for (int k = 0; k < K; ++k) {
          const float4 w4 = *reinterpret_cast<const float4*>(bank + (size_t)k * N + n);
          const float ak = a[k];
          v.x = fmaf(ak, w4.x, v.x); v.y = fmaf(ak, w4.y, v.y); v.z = fmaf(ak, w4.z, v.z); v.w = fmaf(ak, w4.w, v.w);
        }
      }
      float* d = s_w + r * (kPackCols + 1) + 4 * q;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // phase 2: fragment order, one k-step (64 floats) per wave and iteration
    for (int ks = wv; ks < (cols >> 2); ks += 4) {
      const int kc = ks * 4 + (lane >> 4);
      out[((size_t)((c0 >> 2) + ks) * MT + mt) * 64 + lane] = s_w[m * (kPackCols + 1) + kc] * rs;
    }
    __syncthreads();
  }
}

// backward of ctx_pool: dx[b,c,f,t] = dseq[b,f,c]/T + dseq[b,F+t,c]/F  (+ add[b,c,f,t])
__global__ __launch_bounds__(256) void ctx_pool_bwd_kernel(const float* __restrict__ dseq, const float* __restrict__ add,
                                                           float* __restrict__ dx, int C, int F, int T) {
  extern __shared__ float s_g[];                         // [F + T]
  const int plane = blockIdx.x, b = plane / C, c = plane % C;
  const float* g = dseq + (size_t)b * (F + T) * C + c;
  for (int i = threadIdx.x; i < F + T; i += 256) s_g[i] = g[(size_t)i * C] / (float)(i < F ? T : F);
  __syncthreads();
  const size_t base = (size_t)plane * F * T;
  for (int e = threadIdx.x; e < F * T; e += 256) {
    const int f = e / T, t = e - f * T;
    float v = s_g[f] + s_g[F + t];
    if (add) v += add[base + e];
    dx[base + e] = v;
  }
}

// DyReLU-B + coordinate attention on v = a_c z + b_c (BatchNorm affine, a == NULL: v = z):
//   out = max(a1 v + b1, a2 v + b2) * sigmoid(gf[b,f,c]) * sigmoid(gt[b,t,c])
__global__ __launch_bounds__(256) void dyrelu_ca_fwd_kernel(const float* __restrict__ z, const float* __restrict__ a,
                                                            const float* __restrict__ b, const float* __restrict__ coef,
                                                            const float* __restrict__ gf, const float* __restrict__ gt,
                                                            float* __restrict__ out, int C, int Fo, int To) {
  extern __shared__ float s_g[];                         // [Fo + To] sigmoids
  const int plane = blockIdx.x, bb = plane / C, c = plane % C;
  const float av = a ? a[c] : 1.0f, bv = a ? b[c] : 0.0f;
  const float4 cf = *reinterpret_cast<const float4*>(coef + (size_t)plane * 4);
  for (int i = threadIdx.x; i < Fo + To; i += 256) {
    const float g = i < Fo ? gf[((size_t)bb * Fo + i) * C + c] : gt[((size_t)bb * To + (i - Fo)) * C + c];
    s_g[i] = 1.0f / (1.0f + expf(-g));
  }
  __syncthreads();
  const size_t base = (size_t)plane * Fo * To;
  for (int e = threadIdx.x; e < Fo * To; e += 256) {
    const int f = e / To, t = e - f * To;
    const float v = fmaf(av, z[base + e], bv);
    out[base + e] = fmaxf(fmaf(cf.x, v, cf.z), fmaf(cf.y, v, cf.w)) * (s_g[f] * s_g[Fo + t]);
  }
}

// backward: dv (grad w.r.t. v), dcoef (B*C,4), dgf (B,Fo,C), dgt (B,To,C) [pre-sigmoid]
__global__ __launch_bounds__(256) void dyrelu_ca_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                            const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ coef, const float* __restrict__ gf,
                                                            const float* __restrict__ gt, float* __restrict__ dv,
                                                            float* __restrict__ dcoef, float* __restrict__ dgf,
                                                            float* __restrict__ dgt, int C, int Fo, int To) {
  extern __shared__ float s_m[];                         // [Fo + To] sigmoids, [4][To] column partials, [16] coef partials
  float* s_g = s_m;
  float* s_col = s_m + Fo + To;
  float* s_cf = s_col + 4 * To;
  const int plane = blockIdx.x, bb = plane / C, c = plane % C;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float av = a ? a[c] : 1.0f, bv = a ? b[c] : 0.0f;
  const float4 cf = *reinterpret_cast<const float4*>(coef + (size_t)plane * 4);
  for (int i = threadIdx.x; i < Fo + To; i += 256) {
    const float g = i < Fo ? gf[((size_t)bb * Fo + i) * C + c] : gt[((size_t)bb * To + (i - Fo)) * C + c];
    s_g[i] = 1.0f / (1.0f + expf(-g));
  }
  for (int i = threadIdx.x; i < 4 * To; i += 256) s_col[i] = 0.0f;
  __syncthreads();
  const size_t base = (size_t)plane * Fo * To;
  float da1 = 0.f, da2 = 0.f, db1 = 0.f, db2 = 0.f;
  float* mycol = s_col + wv * To;
  for (int f = wv; f < Fo; f += 4) {
    const float af = s_g[f];
    float rs = 0.0f;
    for (int t = lane; t < To; t += 64) {
      const size_t e = base + (size_t)f * To + t;
      const float at = s_g[Fo + t];
      const float v = fmaf(av, z[e], bv);
      const float l1 = fmaf(cf.x, v, cf.z), l2 = fmaf(cf.y, v, cf.w);
      const bool sel = l1 >= l2;
      const float m = sel ? l1 : l2;
      const float d = dout[e];
      const float dm = d * af * at;
      dv[e] = dm * (sel ? cf.x : cf.y);
      if (sel) { da1 = fmaf(dm, v, da1); db1 += dm; } else { da2 = fmaf(dm, v, da2); db2 += dm; }
      rs = fmaf(d * m, at, rs);                 // d out / d af summed over t
      mycol[t] = fmaf(d * m, af, mycol[t]);     // d out / d at summed over this wave's rows
    }
    rs = eat::wave_sum(rs);
    if (lane == 0) dgf[((size_t)bb * Fo + f) * C + c] = rs * af * (1.0f - af);
  }
  da1 = eat::wave_sum(da1); da2 = eat::wave_sum(da2); db1 = eat::wave_sum(db1); db2 = eat::wave_sum(db2);
  if (lane == 0) { s_cf[wv * 4 + 0] = da1; s_cf[wv * 4 + 1] = da2; s_cf[wv * 4 + 2] = db1; s_cf[wv * 4 + 3] = db2; }
  __syncthreads();
  for (int t = threadIdx.x; t < To; t += 256) {
    const float at = s_g[Fo + t];
    dgt[((size_t)bb * To + t) * C + c] = (s_col[t] + s_col[To + t] + s_col[2 * To + t] + s_col[3 * To + t]) * at * (1.0f - at);
  }
  if (threadIdx.x < 4)
    dcoef[(size_t)plane * 4 + threadIdx.x] = s_cf[threadIdx.x] + s_cf[4 + threadIdx.x] + s_cf[8 + threadIdx.x] + s_cf[12 + threadIdx.x];
}

// gradients of the kernel aggregation W_b = sum_k att[b,k] bank[k]:  dbank[k,n] = sum_b att[b,k] G[b,n]
__global__ __launch_bounds__(256) void dyn_dbank_kernel(const float* __restrict__ G, const float* __restrict__ att,
                                                        float* __restrict__ dbank, int B, int K, int N) {
  extern __shared__ float s_att[];                       // [B*K]
  for (int i = threadIdx.x; i < B * K; i += 256) s_att[i] = att[i];
  __syncthreads();
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
      const float g = G[(size_t)b * N + n];
#pragma unroll
      for (int k = 0; k < 8; ++k) if (k < K) acc[k] = fmaf(s_att[b * K + k], g, acc[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < K) dbank[(size_t)k * N + n] = acc[k];
  }
}

// datt[b,k] = <G[b,:], bank[k,:]>   (grid.y = b; partial sums per block, atomics into zeroed datt)
__global__ __launch_bounds__(256) void dyn_datt_kernel(const float* __restrict__ G, const float* __restrict__ bank,
                                                       float* __restrict__ datt, int K, int N) {
  __shared__ float s_red[4][8];
  const int b = blockIdx.y;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float g = G[(size_t)b * N + n];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < K) acc[k] = fmaf(g, bank[(size_t)k * N + n], acc[k]);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = eat::wave_sum(acc[k]);
    if (lane == 0) s_red[wv][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < K)
    atomicAdd(datt + (size_t)b * K + threadIdx.x,
              s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// Both gradients of the aggregation in ONE pass over G (the two kernels above read G twice, each with one load in flight
// per thread): a thread owns 4 consecutive columns n (16-byte loads, 4 samples of them in flight), keeps the K x 4
// dbank accumulators and its bank columns in registers; the K partial dot products of a sample are reduced across the
// wave with DPP adds (scan inside each 16-lane row, row_bcast:15 / :31 across rows; VALU rate, no LDS crossbar), summed
// per block in LDS and added to the zeroed datt once per block.
template <int CTRL, int RM>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, false));
}
__device__ __forceinline__ float wave_total_in_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);   // row_shr:1
  v = dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave total
  return v;
}

template <int K>
__global__ __launch_bounds__(256) void dyn_bank_grad_fused_kernel(const float* __restrict__ G, const float* __restrict__ att,
                                                                  const float* __restrict__ bank, float* __restrict__ dbank,
                                                                  float* __restrict__ datt, int B, int N, int bpb) {
  // blockIdx.y = batch slice of bpb samples (small N: not enough column tiles to fill the chip; dbank is then zeroed by
  // the host and accumulated with atomics)
  extern __shared__ float s_mem[];                       // [B*K] attention, [B*K] per-block datt sums
  float* s_att = s_mem;
  float* s_da = s_mem + B * K;
  for (int i = threadIdx.x; i < B * K; i += 256) { s_att[i] = att[i]; s_da[i] = 0.0f; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t n4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = n4 < (size_t)N;                      // N % 4 == 0 (host)
  const size_t nn = live ? n4 : 0;
  float4 bk[K], acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bk[k] = live ? *reinterpret_cast<const float4*>(bank + (size_t)k * N + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr int U = 4;                                   // samples in flight per thread
  const int b_begin = blockIdx.y * bpb, b_end = b_begin + bpb < B ? b_begin + bpb : B;
  for (int b0 = b_begin; b0 < b_end; b0 += U) {
    float4 g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u < b_end ? b0 + u : b_end - 1;
      g[u] = *reinterpret_cast<const float4*>(G + (size_t)b * N + nn);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u;
      if (b >= b_end) break;                             // uniform
      const float4 gv = live ? g[u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float a = s_att[b * K + k];
        acc[k].x = fmaf(a, gv.x, acc[k].x); acc[k].y = fmaf(a, gv.y, acc[k].y);
        acc[k].z = fmaf(a, gv.z, acc[k].z); acc[k].w = fmaf(a, gv.w, acc[k].w);
        float t = fmaf(gv.x, bk[k].x, fmaf(gv.y, bk[k].y, fmaf(gv.z, bk[k].z, gv.w * bk[k].w)));
        t = wave_total_in_lane63(t);
        if (lane == 63) atomicAdd(&s_da[b * K + k], t);
      }
    }
  }
  if (live) {
    if (gridDim.y == 1) {
#pragma unroll
      for (int k = 0; k < K; ++k) *reinterpret_cast<float4*>(dbank + (size_t)k * N + nn) = acc[k];
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float* d = dbank + (size_t)k * N + nn;
        atomicAdd(d, acc[k].x); atomicAdd(d + 1, acc[k].y); atomicAdd(d + 2, acc[k].z); atomicAdd(d + 3, acc[k].w);
      }
    }
  }
  __syncthreads();
  for (int i = b_begin * K + threadIdx.x; i < b_end * K; i += 256) atomicAdd(datt + i, s_da[i]);
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
  hipLaunchKernelGGL(dyn_pw_pack_kernel, dim3(MT, B), dim3(256), 0, (hipStream_t)stream, bank, att, row_scale, wp, K, Co,
                     Ci, MT);
  return eat::check_launch("eat_dyn_pw_pack");
}

extern "C" int eat_ctx_pool_bwd(const float* dseq, const float* add, float* dx, int B, int C, int F, int T,
                                eat_stream_t stream) {
  eat::clear_stale_error();
  hipLaunchKernelGGL(ctx_pool_bwd_kernel, dim3(B * C), dim3(256), (size_t)(F + T) * sizeof(float), (hipStream_t)stream,
                     dseq, add, dx, C, F, T);
  return eat::check_launch("eat_ctx_pool_bwd");
}

extern "C" int eat_dyrelu_ca_fwd(const float* z, const float* a, const float* b, const float* coef, const float* gate_f,
                                 const float* gate_t, float* out, int B, int C, int Fo, int To, eat_stream_t stream) {
  eat::clear_stale_error();
  hipLaunchKernelGGL(dyrelu_ca_fwd_kernel, dim3(B * C), dim3(256), (size_t)(Fo + To) * sizeof(float),
                     (hipStream_t)stream, z, a, b, coef, gate_f, gate_t, out, C, Fo, To);
  return eat::check_launch("eat_dyrelu_ca_fwd");
}

extern "C" int eat_dyrelu_ca_bwd(const float* dout, const float* z, const float* a, const float* b, const float* coef,
                                 const float* gate_f, const float* gate_t, float* dv, float* dcoef, float* dgate_f,
                                 float* dgate_t, int B, int C, int Fo, int To, eat_stream_t stream) {
  eat::clear_stale_error();
  const size_t smem = (size_t)(Fo + To + 4 * To + 16) * sizeof(float);
  hipLaunchKernelGGL(dyrelu_ca_bwd_kernel, dim3(B * C), dim3(256), smem, (hipStream_t)stream, dout, z, a, b, coef, gate_f,
                     gate_t, dv, dcoef, dgate_f, dgate_t, C, Fo, To);
  return eat::check_launch("eat_dyrelu_ca_bwd");
}

extern "C" int eat_dyn_bank_grad(const float* G, const float* att, const float* bank, float* dbank, float* datt, int B,
                                 int K, int N, eat_stream_t stream) {
  eat::clear_stale_error();
  if (K > 8) return eat::fail(EAT_EINVAL, "eat_dyn_bank_grad: K=%d > 8", K);
  // LDS: the fused kernel stages 2 * B * K floats, the two-pass kernels B * K; both within 48 KB
  if ((size_t)B * K * sizeof(float) > 48 * 1024) return eat::fail(EAT_EINVAL, "eat_dyn_bank_grad: batch too large");
  static const bool two_pass = getenv("EAT_BANK_GRAD_OLD") && atoi(getenv("EAT_BANK_GRAD_OLD")) != 0;
  if (K == 4 && (N & 3) == 0 && !two_pass && (size_t)2 * B * K * sizeof(float) <= 48 * 1024) {            // DynamicConv's k = 4 (models/dymn/dy_block.py:68): one pass over G
    const int gx = (N / 4 + 255) / 256;
    // few column tiles (N <= 64 k): slice the batch over ~512 blocks, a slice walks at least 8 samples (the atomics on
    // dbank cost more than they gain once the column tiles alone fill the chip: 301 k columns 96 vs 105 us)
    int slices = gx <= 64 ? (512 + gx - 1) / gx : 1;
    if (slices > B / 8) slices = B / 8;
    if (slices < 1) slices = 1;
    const int bpb = (B + slices - 1) / slices;
    slices = (B + bpb - 1) / bpb;
    if (slices > 1 && hipMemsetAsync(dbank, 0, (size_t)K * N * sizeof(float), (hipStream_t)stream) != hipSuccess)
      return eat::fail(EAT_ELAUNCH, "eat_dyn_bank_grad: memset failed");
    hipLaunchKernelGGL(dyn_bank_grad_fused_kernel<4>, dim3(gx, slices), dim3(256), (size_t)2 * B * K * sizeof(float),
                       (hipStream_t)stream, G, att, bank, dbank, datt, B, N, bpb);
    return eat::check_launch("eat_dyn_bank_grad");
  }
  int gx = (N + 255) / 256;
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(dyn_dbank_kernel, dim3(gx), dim3(256), (size_t)B * K * sizeof(float), (hipStream_t)stream, G, att,
                     dbank, B, K, N);
  int gy = (N + 256 * 16 - 1) / (256 * 16);
  if (gy > 64) gy = 64;
  if (gy < 1) gy = 1;
  hipLaunchKernelGGL(dyn_datt_kernel, dim3(gy, B), dim3(256), 0, (hipStream_t)stream, G, bank, datt, K, N);
  return eat::check_launch("eat_dyn_bank_grad");
}
