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
#include "act_io.h"

namespace {

// one block per (b,c) plane; wave w takes rows w, w+4, ...
// cm != 0 (round 4): channel-major sequence seq[c][b][l] (a 1x1 conv input of ONE sample with B * (F + T) positions: the
// context generator then runs on the conv / BatchNorm kernels of the feature maps, no transposed copies anywhere)
__global__ __launch_bounds__(256) void ctx_pool_kernel(const float* __restrict__ x, float* __restrict__ seq,
                                                       int C, int F, int T, int cm, int B) {
  extern __shared__ float s_col[];                       // [4][T]
  const int plane = blockIdx.x, b = plane / C, c = plane % C;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* xp = x + (size_t)plane * F * T;
  float* out = cm ? seq + ((size_t)c * B + b) * (F + T) : seq + (size_t)b * (F + T) * C + c;
  const size_t ls = cm ? 1 : (size_t)C;
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
    if (lane == 0) out[(size_t)f * ls] = rs / (float)T;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256)
    out[(size_t)(F + t) * ls] = (s_col[t] + s_col[T + t] + s_col[2 * T + t] + s_col[3 * T + t]) / (float)F;
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
                                                          int K, int Co, int Ci, int MT, int trans) {
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
    if (trans) {
      // the bank stores W_k^T ((Ci, Co) row-major): packed row m = stored column, 16 consecutive m are one 64-byte run
      for (int e = tid; e < 4 * cols; e += 256) {
        const int col = e >> 2, q = e & 3;                          // (packed column, group of 4 packed rows)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * q < rows) {                                         // host: Co % 4 == 0
          const size_t n = (size_t)(c0 + col) * Co + m0 + 4 * q;
          for (int k = 0; k < K; ++k) {
            const float4 w4 = *reinterpret_cast<const float4*>(bank + (size_t)k * N + n);
            const float ak = a[k];
            v.x = fmaf(ak, w4.x, v.x); v.y = fmaf(ak, w4.y, v.y); v.z = fmaf(ak, w4.z, v.z); v.w = fmaf(ak, w4.w, v.w);
          }
        }
        s_w[(4 * q + 0) * (kPackCols + 1) + col] = v.x; s_w[(4 * q + 1) * (kPackCols + 1) + col] = v.y;
        s_w[(4 * q + 2) * (kPackCols + 1) + col] = v.z; s_w[(4 * q + 3) * (kPackCols + 1) + col] = v.w;
      }
    } else
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

// The same aggregation as bf16 hi / lo fragments of v_mfma_f32_16x16x32_bf16 (eat_pw_prepack_bf16's layout, one pack per
// sample):  wp[b][((kk*MT + mt)*2 + h)*512 + lane*8 + i] = part h of W_b[mt*16 + (lane&15)][kk*32 + 8*(lane>>4) + i].
// Block = (16-row m-tile, sample); 16 x 256 aggregated weights through LDS as above, then every lane writes its 8
// consecutive k of one (kk, h) fragment as ONE 16-byte store (a wave = one 1 KiB fragment).
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
// Round 6: a block packs SB consecutive SAMPLES of its m-tile.  With K == KT (= 4, every DynamicConv of the reference) the K bank
// tiles of a 256-column piece are loaded ONCE into registers and mixed with each sample's attention in turn - before, every
// (m-tile, sample) block re-read the K x 16 x 256 floats through L2 (8 x the bytes it wrote: 1.0 - 1.8 TB/s of packed output).
// KT == 0: any K, the bank values are fetched per sample as before.
template <int KT>
__global__ __launch_bounds__(256) void dyn_pw_pack_bf16_kernel(const float* __restrict__ bank, const float* __restrict__ att,
                                                               __bf16* __restrict__ wp, int K, int Co, int Ci, int MT, int KK,
                                                               int trans, int np2, int B, int SB) {
  // np2 = 2: hi / lo fragments (bf16x3); np2 = 1: the hi part only (plain bf16 operands, the bf16-storage plan)
  __shared__ float s_w[16 * (kPackCols + 1)];
  const int mt = blockIdx.x, tid = threadIdx.x;
  const int b0 = blockIdx.y * SB;
  const int nb = (B - b0) < SB ? (B - b0) : SB;
  const int m0 = mt * 16;
  const int rows = (Co - m0) < 16 ? (Co - m0) : 16;
  const size_t N = (size_t)Co * Ci;
  const int lane = tid & 63, wv = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  constexpr int KR = KT > 0 ? KT : 1;
  {
    const int c0 = blockIdx.z * kPackCols;                             // the block's 256-column piece
    const int cols = (Ci - c0) < kPackCols ? (Ci - c0) : kPackCols;     // multiple of 4
    // the thread's 4 float4 positions of the piece: offset into a bank (or -1: outside the matrix -> zeros) and LDS slot
    long long noff[4];
    int slot[4];
    float4 wreg[4][KR];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + 256 * it;
      if (trans) {                                                     // 4 * kPackCols = 1024 (col, 4-row group) positions
        const int col = e >> 2, q = e & 3;
        noff[it] = (4 * q < rows && col < cols) ? (long long)((size_t)(c0 + col) * Co + m0 + 4 * q) : -1;
        slot[it] = (4 * q) * (kPackCols + 1) + col;
      } else {                                                         // 16 * (kPackCols / 4) = 1024 (row, 4-col group) positions
        const int r = e / (kPackCols >> 2), q = e - r * (kPackCols >> 2);
        noff[it] = (r < rows && 4 * q < cols) ? (long long)((size_t)(m0 + r) * Ci + c0 + 4 * q) : -1;
        slot[it] = r * (kPackCols + 1) + 4 * q;
      }
      if constexpr (KT > 0) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
          wreg[it][k] = noff[it] >= 0 ? *reinterpret_cast<const float4*>(bank + (size_t)k * N + noff[it]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    for (int sb = 0; sb < nb; ++sb) {
      const float* a = att + (size_t)(b0 + sb) * K;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (KT > 0) {
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            const float ak = a[k];
            const float4 w4 = wreg[it][k];
            v.x = fmaf(ak, w4.x, v.x); v.y = fmaf(ak, w4.y, v.y); v.z = fmaf(ak, w4.z, v.z); v.w = fmaf(ak, w4.w, v.w);
          }
        } else if (noff[it] >= 0) {
          for (int k = 0; k < K; ++k) {
            const float4 w4 = *reinterpret_cast<const float4*>(bank + (size_t)k * N + noff[it]);
            const float ak = a[k];
            v.x = fmaf(ak, w4.x, v.x); v.y = fmaf(ak, w4.y, v.y); v.z = fmaf(ak, w4.z, v.z); v.w = fmaf(ak, w4.w, v.w);
          }
        }
        float* d = s_w + slot[it];
        if (trans) {                                                   // the float4 runs down 4 ROWS of one column
          d[0] = v.x; d[kPackCols + 1] = v.y; d[2 * (kPackCols + 1)] = v.z; d[3 * (kPackCols + 1)] = v.w;
        } else {                                                       // ... along 4 columns of one row (zeros beyond Ci / Co)
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      }
      __syncthreads();
      __bf16* out = wp + (size_t)(b0 + sb) * KK * MT * 512 * np2;
      const int nkk = (cols + 31) >> 5;
      for (int j = wv; j < nkk * np2; j += 4) {                      // (32-column chunk, hi / lo) fragments of this piece
        const int kl = np2 == 2 ? j >> 1 : j, h = np2 == 2 ? j & 1 : 0;
        bf16x8_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = s_w[m * (kPackCols + 1) + kl * 32 + 8 * kq + i];
          const __bf16 hi = (__bf16)v;
          o[i] = h ? (__bf16)(v - (float)hi) : hi;
        }
        *reinterpret_cast<bf16x8_t*>(out + (((size_t)((c0 >> 5) + kl) * MT + mt) * np2 + h) * 512 + lane * 8) = o;
      }
      __syncthreads();
    }
  }
}

// backward of ctx_pool: dx[b,c,f,t] = dseq[b,f,c]/T + dseq[b,F+t,c]/F  (+ add[b,c,f,t])
__global__ __launch_bounds__(256) void ctx_pool_bwd_kernel(const float* __restrict__ dseq, const float* __restrict__ add,
                                                           float* __restrict__ dx, int C, int F, int T, int cm, int B) {
  extern __shared__ float s_g[];                         // [F + T]
  const int plane = blockIdx.x, b = plane / C, c = plane % C;
  const float* g = cm ? dseq + ((size_t)c * B + b) * (F + T) : dseq + (size_t)b * (F + T) * C + c;
  const size_t ls = cm ? 1 : (size_t)C;
  for (int i = threadIdx.x; i < F + T; i += 256) s_g[i] = g[(size_t)i * ls] / (float)(i < F ? T : F);
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
  constexpr int U = 4;                                   // samples in flight per thread (8 measured the same: 2.03 vs 2.06 ms per dymn20 step)
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


// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the context generator and DyReLU-B * CoordAtt of the training step, second form.  The kernels above keep the
// context sequence position-major ((B, L, C): every plane of the DyReLU kernels gathers its Fo + To gate values with
// stride C - on the 4 x 32 planes of the last stage the 36 gathered cache lines outweigh the plane's own 512 bytes, 1 TB/s
// - and every Linear backward of the context path needs transposed copies of its operands) and give one 256-thread block
// to every plane.  Here
//   * the context sequence is channel-major with the batch folded into the position axis, seq (C, B, F + T): joint_conv /
//     conv_f / conv_t are 1x1 convs of ONE sample of B * L positions (B * L is a multiple of 4 whenever B is: no padding)
//     on the MFMA conv kernels, joint_norm is the BatchNorm of the feature maps, their backward the conv backward - no
//     transposed copy anywhere; the gates come out as (C, B, Fo) / (C, B, To): a plane's gate row is one contiguous run;
//   * a wave owns a whole plane (two planes for <= 32 columns): lane l holds columns l, l + LPP, ... of every row -
//     coalesced row segments, the column sums of the backward stay in registers, row sums / coefficient sums /
//     BatchNorm sums are lane-group reductions; the sigmoids are evaluated on load, their derivative on store;
//   * the backward also emits the two per-plane sums the BatchNorm backward of depth_norm needs (sum dv, sum dv * z): its
//     reduce pass over (dv, z) disappears (dyn_bn_bwd_combine_kernel turns them into the fp64 channel sums).
// Reference: models/dymn/dy_block.py:172-188 (DyReLU-B), :195-201 (CoordAtt), :399-403 (order inside DY_Block).

// ctx_split: the context sequence after joint_norm + Hardswish, g (H, B, L = F + T) channel-major, is cut into the inputs
// of conv_f / conv_t (models/dymn/dy_block.py:240-252): h_cf (H, B, Fo) = [AvgPool(3, stride, pad 1) of] g[.., :F],
// h_ct (H, B, To) likewise of g[.., F:], and h_c (B, H) = mean over L (:244).  One wave per (h, b) row.
__device__ __forceinline__ float pool3_at(const float* __restrict__ r, int i, int n, int stride) {
  if (stride == 1) return r[i];
  const int j = 2 * i;
  return ((j - 1 >= 0 ? r[j - 1] : 0.0f) + r[j] + (j + 1 < n ? r[j + 1] : 0.0f)) * (1.0f / 3.0f);
}
__global__ __launch_bounds__(256) void ctx_split_kernel(const float* __restrict__ g, float* __restrict__ hcf,
                                                        float* __restrict__ hct, float* __restrict__ hc, int H, int B, int F,
                                                        int T, int Fo, int To, int stride) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= H * B) return;
  const int h = row / B, b = row - h * B;
  const float* r = g + (size_t)row * (F + T);
  float s = 0.0f;
  for (int l = lane; l < F + T; l += 64) s += r[l];
  s = eat::wave_sum(s);
  if (lane == 0) hc[(size_t)b * H + h] = s / (float)(F + T);
  for (int i = lane; i < Fo; i += 64) hcf[(size_t)row * Fo + i] = pool3_at(r, i, F, stride);
  for (int i = lane; i < To; i += 64) hct[(size_t)row * To + i] = pool3_at(r + F, i, T, stride);
}
// backward: dg[h][b][l] = dhc[b][h] / L + pool^T(dhcf)[l] (l < F) or pool^T(dhct)[l - F]
__device__ __forceinline__ float pool3_t_at(const float* __restrict__ d, int j, int no, int stride) {
  if (stride == 1) return d[j];
  if ((j & 1) == 0) return (j >> 1) < no ? d[j >> 1] * (1.0f / 3.0f) : 0.0f;
  const int i0 = (j - 1) >> 1, i1 = (j + 1) >> 1;
  return ((i0 < no ? d[i0] : 0.0f) + (i1 < no ? d[i1] : 0.0f)) * (1.0f / 3.0f);
}
__global__ __launch_bounds__(256) void ctx_split_bwd_kernel(const float* __restrict__ dhcf, const float* __restrict__ dhct,
                                                            const float* __restrict__ dhc, float* __restrict__ dg, int H, int B,
                                                            int F, int T, int Fo, int To, int stride) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= H * B) return;
  const int h = row / B, b = row - h * B;
  const float m = dhc ? dhc[(size_t)b * H + h] / (float)(F + T) : 0.0f;
  float* o = dg + (size_t)row * (F + T);
  for (int j = lane; j < F; j += 64) o[j] = m + pool3_t_at(dhcf + (size_t)row * Fo, j, Fo, stride);
  for (int j = lane; j < T; j += 64) o[F + j] = m + pool3_t_at(dhct + (size_t)row * To, j, To, stride);
}

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LPP lanes per plane (64 / LPP planes of consecutive channels per wave), NC column slots per lane; a slot is CW columns wide:
// one for fp32 storage, TWO ADJACENT columns for bf16 storage (one 4-byte access per slot: 2-byte accesses ran the bf16 kernels
// at the instruction count of the fp32 ones, i.e. at half their bandwidth).  To <= LPP * NC * CW.
// A bf16 row of odd width starts on a 2-byte boundary on every other row: the slot access is ONE dword at a 2-byte aligned
// address (gfx9 global memory handles it); the last slot of an odd-width row owns one column only - its load is moved back by
// one element and takes the high half (so that it never reaches past the tensor), its store is a 2-byte store.
template <typename ST> struct DyIo;
template <> struct DyIo<float> {
  static constexpr int CW = 1;
  static __device__ __forceinline__ void ld(const float* p, bool, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void st(float* p, bool, bool ok0, const float (&v)[1]) { if (ok0) *p = v[0]; }
};
template <> struct DyIo<eat::bf16_t> {
  static constexpr int CW = 2;
  // p: address of the slot's first column; part: the second column does not exist
  static __device__ __forceinline__ void ld(const eat::bf16_t* p, bool part, float (&v)[2]) {
    const unsigned w = *reinterpret_cast<const unsigned*>(part ? p - 1 : p);
    v[0] = part ? eat::bf_hi(w) : eat::bf_lo(w);
    v[1] = part ? 0.0f : eat::bf_hi(w);
  }
  static __device__ __forceinline__ void st(eat::bf16_t* p, bool part, bool ok0, const float (&v)[2]) {
    if (!ok0) return;
    if (part) *p = (eat::bf16_t)v[0];
    else *reinterpret_cast<unsigned*>(p) = eat::pack_bf2(v[0], v[1]);
  }
};

// ST: storage type of the feature maps z / out (act_io.h; bf16 in the bf16-storage plan - the output is rounded on store)
template <int LPP, int NC, typename ST = float>
__global__ __launch_bounds__(256) void dyrelu_ca_fwd2_kernel(const ST* __restrict__ z, const float* __restrict__ a,
                                                             const float* __restrict__ b, const float* __restrict__ coef,
                                                             const float* __restrict__ gf, const float* __restrict__ gt,
                                                             ST* __restrict__ out, int n_planes, int C, int Fo, int To) {
  constexpr int NPW = 64 / LPP, CW = DyIo<ST>::CW;
  const int lane = threadIdx.x & 63, l = lane & (LPP - 1);
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  int plane = wave * NPW + lane / LPP;
  const bool mine = plane < n_planes;
  if (!mine) plane = n_planes - 1;
  const int c = plane % C;
  const float av = a ? a[c] : 1.0f, bv = a ? b[c] : 0.0f;
  const float4 cf = *reinterpret_cast<const float4*>(coef + (size_t)plane * 4);
  const int Bn = n_planes / C;
  const size_t grow = (size_t)c * Bn + plane / C;                  // gate row of plane (b, c) in the (C, B, .) tables
  const float* gfp = gf + grow * Fo;
  const float* gtp = gt + grow * To;
  const ST* zp = z + (size_t)plane * Fo * To;
  ST* op = out + (size_t)plane * Fo * To;
  float at[NC][CW];
  bool ok[NC], part[NC];
  int tc[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int t = CW * (l + LPP * j);
    ok[j] = mine && t < To;
    part[j] = CW == 2 && t + 1 >= To;
    tc[j] = t < To ? t : 0;
#pragma unroll
    for (int q = 0; q < CW; ++q) at[j][q] = sigm(gtp[t + q < To ? t + q : 0]);
  }
  constexpr int RU = NC * CW >= 4 ? 2 : 4;                         // rows in flight
  for (int f0 = 0; f0 < Fo; f0 += RU) {
    float v[RU][NC][CW], af[RU];
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      const int f = f0 + r < Fo ? f0 + r : Fo - 1;
      af[r] = sigm(gfp[f]);
#pragma unroll
      for (int j = 0; j < NC; ++j) DyIo<ST>::ld(zp + (size_t)f * To + tc[j], part[j] && CW * (l + LPP * j) < To, v[r][j]);
    }
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      if (f0 + r >= Fo) break;                                      // wave-uniform
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        float o[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) {
          const float u = fmaf(av, v[r][j][q], bv);
          o[q] = fmaxf(fmaf(cf.x, u, cf.z), fmaf(cf.y, u, cf.w)) * (af[r] * at[j][q]);
        }
        DyIo<ST>::st(op + (size_t)(f0 + r) * To + tc[j], part[j], ok[j], o);
      }
    }
  }
}

// ST: storage type of dout / z / dv; the BatchNorm-backward partials are those of dv AS STORED
template <int LPP, int NC, typename ST = float>
__global__ __launch_bounds__(256) void dyrelu_ca_bwd2_kernel(const ST* __restrict__ dout, const ST* __restrict__ z,
                                                             const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ coef, const float* __restrict__ gf,
                                                             const float* __restrict__ gt, ST* __restrict__ dv,
                                                             float* __restrict__ dcoef, float* __restrict__ dgf,
                                                             float* __restrict__ dgt, float* __restrict__ bnpart,
                                                             int n_planes, int C, int Fo, int To) {
  constexpr int NPW = 64 / LPP, CW = DyIo<ST>::CW;
  const int lane = threadIdx.x & 63, l = lane & (LPP - 1);
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  int plane = wave * NPW + lane / LPP;
  const bool mine = plane < n_planes;
  if (!mine) plane = n_planes - 1;
  const int c = plane % C;
  const float av = a ? a[c] : 1.0f, bv = a ? b[c] : 0.0f;
  const float4 cf = *reinterpret_cast<const float4*>(coef + (size_t)plane * 4);
  const int Bn = n_planes / C;
  const size_t grow = (size_t)c * Bn + plane / C;
  const float* gfp = gf + grow * Fo;
  const float* gtp = gt + grow * To;
  float* dgfp = dgf + grow * Fo;
  float* dgtp = dgt + grow * To;
  const size_t base = (size_t)plane * Fo * To;
  float at[NC][CW], cs[NC][CW];
  bool ok[NC][CW], part[NC];
  int tc[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int t = CW * (l + LPP * j);
    part[j] = CW == 2 && t + 1 >= To;
    tc[j] = t < To ? t : 0;
#pragma unroll
    for (int q = 0; q < CW; ++q) {
      ok[j][q] = t + q < To;
      at[j][q] = sigm(gtp[t + q < To ? t + q : 0]);
      cs[j][q] = 0.0f;
    }
  }
  float da1 = 0.f, da2 = 0.f, db1 = 0.f, db2 = 0.f, s1 = 0.f, s2 = 0.f;
  constexpr int RU = NC * CW >= 4 ? 2 : 4;
  for (int f0 = 0; f0 < Fo; f0 += RU) {
    float zv[RU][NC][CW], dd[RU][NC][CW], af[RU];
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      const int f = f0 + r < Fo ? f0 + r : Fo - 1;
      af[r] = sigm(gfp[f]);
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const size_t e = base + (size_t)f * To + tc[j];
        const bool pt = part[j] && ok[j][0];
        DyIo<ST>::ld(z + e, pt, zv[r][j]);
        DyIo<ST>::ld(dout + e, pt, dd[r][j]);
      }
    }
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      if (f0 + r >= Fo) break;                                      // wave-uniform
      float rs = 0.0f;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        float g[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) {
          const float zr = zv[r][j][q];
          const float u = fmaf(av, zr, bv);
          const float l1 = fmaf(cf.x, u, cf.z), l2 = fmaf(cf.y, u, cf.w);
          const bool sel = l1 >= l2;
          const float m = sel ? l1 : l2;
          const float d = ok[j][q] ? dd[r][j][q] : 0.0f;
          const float dm = d * (af[r] * at[j][q]);
          g[q] = eat::Io<ST>::rnd(dm * (sel ? cf.x : cf.y));
          const float dmv = dm * u;
          da1 += sel ? dmv : 0.0f; db1 += sel ? dm : 0.0f;
          da2 += sel ? 0.0f : dmv; db2 += sel ? 0.0f : dm;
          s1 += g[q];
          s2 = fmaf(g[q], zr, s2);
          const float dmm = d * m;
          rs = fmaf(dmm, at[j][q], rs);
          cs[j][q] = fmaf(dmm, af[r], cs[j][q]);
        }
        DyIo<ST>::st(dv + base + (size_t)(f0 + r) * To + tc[j], part[j], ok[j][0] && mine, g);
      }
      rs = group_sum<LPP>(rs);
      if (l == 0 && mine) dgfp[f0 + r] = rs * af[r] * (1.0f - af[r]);       // gradient w.r.t. the PRE-sigmoid gate
    }
  }
  if (mine) {
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
      for (int q = 0; q < CW; ++q)
        if (ok[j][q]) dgtp[CW * (l + LPP * j) + q] = cs[j][q] * at[j][q] * (1.0f - at[j][q]);
  }
  da1 = group_sum<LPP>(da1); da2 = group_sum<LPP>(da2); db1 = group_sum<LPP>(db1); db2 = group_sum<LPP>(db2);
  s1 = group_sum<LPP>(s1); s2 = group_sum<LPP>(s2);
  if (l == 0 && mine) {
    *reinterpret_cast<float4*>(dcoef + (size_t)plane * 4) = make_float4(da1, da2, db1, db2);
    if (bnpart) { bnpart[(size_t)plane * 2] = s1; bnpart[(size_t)plane * 2 + 1] = s2; }
  }
}

// sums[c] = sum_b part[b][c][0], sums[C + c] = invstd[c] * (sum_b part[b][c][1] - mean[c] * sums[c])   (fp64): the channel
// sums (sum dv, sum dv * xhat) of the BatchNorm backward from the per-plane (sum dv, sum dv * z) of dyrelu_ca_bwd2 /
// the merged depthwise backward; parts may carry `inner` slots per plane: part0[(b*C + c)*inner + i] (two arrays)
__global__ __launch_bounds__(256) void dyn_bn_bwd_combine_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                                 int stride_e, int B, int C, int inner,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 double* __restrict__ sums, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int i = threadIdx.x; i < B * inner; i += 256) {
    const int bb = i / inner, q = i - bb * inner;
    const size_t e = (((size_t)bb * C + c) * inner + q) * stride_e;
    a0 += (double)p0[e];
    a1 += (double)p1[e];
  }
  __shared__ double s_r[2][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
  if (lane == 0) { s_r[0][wv] = a0; s_r[1][wv] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t0 = (s_r[0][0] + s_r[0][1]) + (s_r[0][2] + s_r[0][3]);
    const double t1 = (s_r[1][0] + s_r[1][1]) + (s_r[1][2] + s_r[1][3]);
    const double gx = (double)invstd[c] * (t1 - (double)mean[c] * t0);
    sums[c] = t0;
    sums[C + c] = gx;
    dbeta[c] = (float)t0;
    dgamma[c] = (float)gx;
  }
}

}  // namespace

extern "C" int eat_ctx_pool(const float* x, float* seq, int B, int C, int F, int T, eat_stream_t stream) {
  eat::clear_stale_error();
  const size_t smem = (size_t)4 * T * sizeof(float);
  if (smem > 64 * 1024) return eat::fail(EAT_EINVAL, "eat_ctx_pool: T=%d too wide", T);
  hipLaunchKernelGGL(ctx_pool_kernel, dim3(B * C), dim3(256), smem, (hipStream_t)stream, x, seq, C, F, T, 0, B);
  return eat::check_launch("eat_ctx_pool");
}

// channel-major form: seq (C, B, F+T) (see the kernel)
extern "C" int eat_ctx_pool_cm(const float* x, float* seq, int B, int C, int F, int T, eat_stream_t stream) {
  eat::clear_stale_error();
  const size_t smem = (size_t)4 * T * sizeof(float);
  if (smem > 64 * 1024) return eat::fail(EAT_EINVAL, "eat_ctx_pool_cm: T=%d too wide", T);
  hipLaunchKernelGGL(ctx_pool_kernel, dim3(B * C), dim3(256), smem, (hipStream_t)stream, x, seq, C, F, T, 1, B);
  return eat::check_launch("eat_ctx_pool_cm");
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

static int dyn_pw_pack_impl(const float* bank, const float* att, const float* row_scale, float* wp, int B, int K, int Co,
                            int Ci, int trans, eat_stream_t stream) {
  eat::clear_stale_error();
  if (Ci % 4 != 0) return eat::fail(EAT_EINVAL, "eat_dyn_pw_pack: Ci=%d must be a multiple of 4", Ci);
  if (trans && Co % 4 != 0) return eat::fail(EAT_EINVAL, "eat_dyn_pw_pack_t: Co=%d must be a multiple of 4", Co);
  const int MT = (Co + 15) / 16;
  hipLaunchKernelGGL(dyn_pw_pack_kernel, dim3(MT, B), dim3(256), 0, (hipStream_t)stream, bank, att, row_scale, wp, K, Co,
                     Ci, MT, trans);
  return eat::check_launch("eat_dyn_pw_pack");
}

extern "C" int eat_dyn_pw_pack(const float* bank, const float* att, const float* row_scale, float* wp, int B, int K,
                               int Co, int Ci, eat_stream_t stream) {
  return dyn_pw_pack_impl(bank, att, row_scale, wp, B, K, Co, Ci, 0, stream);
}

// ... from the bank of the TRANSPOSED matrices: bank_t (K, Ci*Co) holds W_k^T row-major, i.e. the parameter of the conv
// whose data gradient this is - packs sum_k att[b,k] W_k (Co x Ci) without a transposed copy of the bank
extern "C" int eat_dyn_pw_pack_t(const float* bank_t, const float* att, const float* row_scale, float* wp, int B, int K,
                                 int Co, int Ci, eat_stream_t stream) {
  return dyn_pw_pack_impl(bank_t, att, row_scale, wp, B, K, Co, Ci, 1, stream);
}

static int dyn_pw_pack_bf16_impl(const float* bank, const float* att, void* wp, int B, int K, int Co, int Ci, int trans,
                                 eat_stream_t stream, int np2 = 2) {
  eat::clear_stale_error();
  if (Ci % 4 != 0) return eat::fail(EAT_EINVAL, "eat_dyn_pw_pack_bf16: Ci=%d must be a multiple of 4", Ci);
  if (trans && Co % 4 != 0) return eat::fail(EAT_EINVAL, "eat_dyn_pw_pack_bf16_t: Co=%d must be a multiple of 4", Co);
  if (B < 1 || K < 1 || Co < 1) return eat::fail(EAT_EINVAL, "eat_dyn_pw_pack_bf16: bad shape");
  const int MT = (Co + 15) / 16, KK = (Ci + 31) / 32;
  // a block = (m-tile, SB samples, 256-column piece); SB: as many samples as keep >= ~2048 blocks (8 per CU) in the launch, at most 8
  const int pieces = (Ci + kPackCols - 1) / kPackCols;
  int SB = 8;
  while (SB > 1 && (long long)MT * pieces * ((B + SB - 1) / SB) < 2048) SB >>= 1;
  const dim3 grid(MT, (B + SB - 1) / SB, pieces);
  if (K == 4)
    hipLaunchKernelGGL(dyn_pw_pack_bf16_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, bank, att,
                       reinterpret_cast<__bf16*>(wp), K, Co, Ci, MT, KK, trans, np2, B, SB);
  else
    hipLaunchKernelGGL(dyn_pw_pack_bf16_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, bank, att,
                       reinterpret_cast<__bf16*>(wp), K, Co, Ci, MT, KK, trans, np2, B, SB);
  return eat::check_launch("eat_dyn_pw_pack_bf16");
}

extern "C" int eat_dyn_pw_pack_bf16(const float* bank, const float* att, void* wp, int B, int K, int Co, int Ci,
                                    eat_stream_t stream) {
  return dyn_pw_pack_bf16_impl(bank, att, wp, B, K, Co, Ci, 0, stream);
}

extern "C" int eat_dyn_pw_pack_bf16_t(const float* bank_t, const float* att, void* wp, int B, int K, int Co, int Ci,
                                      eat_stream_t stream) {
  return dyn_pw_pack_bf16_impl(bank_t, att, wp, B, K, Co, Ci, 1, stream);
}

// plain bf16 packs (one fragment per (k-chunk, m-tile), eat_pw_prepack_bf16's layout with split = 0) for the per-sample convs of
// the bf16-storage plan (eat_pw_conv_dyn_b16_fwd); trans != 0: from the bank of the transposed matrices (the data gradient)
extern "C" int eat_dyn_pw_pack_b16(const float* bank, const float* att, void* wp, int B, int K, int Co, int Ci, int trans,
                                   eat_stream_t stream) {
  return dyn_pw_pack_bf16_impl(bank, att, wp, B, K, Co, Ci, trans ? 1 : 0, stream, 1);
}

extern "C" int eat_ctx_pool_bwd(const float* dseq, const float* add, float* dx, int B, int C, int F, int T,
                                eat_stream_t stream) {
  eat::clear_stale_error();
  hipLaunchKernelGGL(ctx_pool_bwd_kernel, dim3(B * C), dim3(256), (size_t)(F + T) * sizeof(float), (hipStream_t)stream,
                     dseq, add, dx, C, F, T, 0, B);
  return eat::check_launch("eat_ctx_pool_bwd");
}

extern "C" int eat_ctx_pool_cm_bwd(const float* dseq, const float* add, float* dx, int B, int C, int F, int T,
                                   eat_stream_t stream) {
  eat::clear_stale_error();
  hipLaunchKernelGGL(ctx_pool_bwd_kernel, dim3(B * C), dim3(256), (size_t)(F + T) * sizeof(float), (hipStream_t)stream,
                     dseq, add, dx, C, F, T, 1, B);
  return eat::check_launch("eat_ctx_pool_cm_bwd");
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

namespace {
__global__ __launch_bounds__(256) void zero_floats_kernel(float* __restrict__ p, int n4) {     // n4 float4 (N % 4 == 0 on this path)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n4) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace

extern "C" int eat_dyn_bank_grad(const float* G, const float* att, const float* bank, float* dbank, float* datt, int B,
                                 int K, int N, eat_stream_t stream) {
  eat::clear_stale_error();
  if (K > 8) return eat::fail(EAT_EINVAL, "eat_dyn_bank_grad: K=%d > 8", K);
  // LDS: the fused kernel stages 2 * B * K floats, the two-pass kernels B * K; both within 48 KB
  if ((size_t)B * K * sizeof(float) > 48 * 1024) return eat::fail(EAT_EINVAL, "eat_dyn_bank_grad: batch too large");
  if (K == 4 && (N & 3) == 0 && (size_t)2 * B * K * sizeof(float) <= 48 * 1024) {            // DynamicConv's k = 4 (models/dymn/dy_block.py:68): one pass over G
    const int gx = (N / 4 + 255) / 256;
    // few column tiles (N <= 64 k): slice the batch over ~512 blocks, a slice walks at least 8 samples (the atomics on
    // dbank cost more than they gain once the column tiles alone fill the chip: 301 k columns 96 vs 105 us)
    int slices = gx <= 64 ? (512 + gx - 1) / gx : 1;
    if (slices > B / 8) slices = B / 8;
    if (slices < 1) slices = 1;
    const int bpb = (B + slices - 1) / slices;
    slices = (B + bpb - 1) / bpb;
    // (a KERNEL, not hipMemsetAsync: inside a captured hipGraph the memset node of ROCm 7.0 was not ordered reliably against the
    //  kernel nodes around it - with every torch.empty poisoned with NaN the second replay of the dymn20 step left a quarter of
    //  these small dbank tensors non-finite, and unpoisoned runs showed a NaN loss in ~1 of 20 fresh processes)
    if (slices > 1)
      hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)((K * N / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dbank, K * N / 4);
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

// ---- round 4: channel-major gate table + one-wave-per-plane DyReLU-B * CoordAtt (see the kernels)
extern "C" int eat_ctx_split(const float* g, float* hcf, float* hct, float* hc, int H, int B, int F, int T, int stride,
                             eat_stream_t stream) {
  eat::clear_stale_error();
  if (stride != 1 && stride != 2) return eat::fail(EAT_EINVAL, "eat_ctx_split: stride %d", stride);
  const int Fo = (F - 1) / stride + 1, To = (T - 1) / stride + 1;
  hipLaunchKernelGGL(ctx_split_kernel, dim3((H * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, hcf, hct, hc, H, B, F, T, Fo,
                     To, stride);
  return eat::check_launch("eat_ctx_split");
}

extern "C" int eat_ctx_split_bwd(const float* dhcf, const float* dhct, const float* dhc, float* dg, int H, int B, int F, int T,
                                 int stride, eat_stream_t stream) {
  eat::clear_stale_error();
  if (stride != 1 && stride != 2) return eat::fail(EAT_EINVAL, "eat_ctx_split_bwd: stride %d", stride);
  const int Fo = (F - 1) / stride + 1, To = (T - 1) / stride + 1;
  hipLaunchKernelGGL(ctx_split_bwd_kernel, dim3((H * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, dhcf, dhct, dhc, dg, H, B,
                     F, T, Fo, To, stride);
  return eat::check_launch("eat_ctx_split_bwd");
}

// (To <= LPP * NC * CW columns per lane group: CW = 1 for fp32 storage, 2 for bf16 - see DyIo)
#define EAT_DYRELU2_LAUNCH(KERNEL, ST, LPP_, NC_, ...)                                                                 \
  hipLaunchKernelGGL((KERNEL<LPP_, NC_, ST>), dim3(((n_planes + (64 / LPP_) - 1) / (64 / LPP_) + 3) / 4), dim3(256), 0, hs, __VA_ARGS__)
#define EAT_DYRELU2_DISPATCH(KERNEL, ST, ...)                                                                         \
  do {                                                                                                                \
    const int n_planes = B * C;                                                                                       \
    const int tw = (To + DyIo<ST>::CW - 1) / DyIo<ST>::CW;               /* slots per row */                           \
    if (tw <= 16) EAT_DYRELU2_LAUNCH(KERNEL, ST, 16, 1, __VA_ARGS__);                                                  \
    else if (tw <= 32) EAT_DYRELU2_LAUNCH(KERNEL, ST, 32, 1, __VA_ARGS__);                                             \
    else if (tw <= 64) EAT_DYRELU2_LAUNCH(KERNEL, ST, 64, 1, __VA_ARGS__);                                             \
    else if (tw <= 128) EAT_DYRELU2_LAUNCH(KERNEL, ST, 64, 2, __VA_ARGS__);                                            \
    else if (tw <= 256) EAT_DYRELU2_LAUNCH(KERNEL, ST, 64, 4, __VA_ARGS__);                                            \
    else EAT_DYRELU2_LAUNCH(KERNEL, ST, 64, 8, __VA_ARGS__);                                                           \
  } while (0)

// out = max(a1 v + b1, a2 v + b2) * sigmoid(gate_f[c,b,f]) * sigmoid(gate_t[c,b,t]), v = a_c z + b_c (a, b NULL: v = z);
// gates channel-major (C, B, Fo) / (C, B, To), pre-sigmoid: the outputs of conv_f / conv_t on the channel-major context
extern "C" int eat_dyrelu_ca_fwd2(const float* z, const float* a, const float* b, const float* coef, const float* gate_f,
                                  const float* gate_t, float* out, int B, int C, int Fo, int To, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || Fo < 1 || To < 1 || To > 512) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_fwd2: bad shape (To <= 512)");
  if ((a == nullptr) != (b == nullptr)) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_fwd2: a and b come together");
  hipStream_t hs = (hipStream_t)stream;
  EAT_DYRELU2_DISPATCH(dyrelu_ca_fwd2_kernel, float, z, a, b, coef, gate_f, gate_t, out, n_planes, C, Fo, To);
  return eat::check_launch("eat_dyrelu_ca_fwd2");
}

// bf16-storage twin (act_io.h): z and out are bf16 in HBM
extern "C" int eat_dyrelu_ca_fwd2_b16(const void* z, const float* a, const float* b, const float* coef, const float* gate_f,
                                      const float* gate_t, void* out, int B, int C, int Fo, int To, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || Fo < 1 || To < 1 || To > 512) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_fwd2_b16: bad shape (To <= 512)");
  if ((a == nullptr) != (b == nullptr)) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_fwd2_b16: a and b come together");
  hipStream_t hs = (hipStream_t)stream;
  const eat::bf16_t* z16 = reinterpret_cast<const eat::bf16_t*>(z);
  eat::bf16_t* o16 = reinterpret_cast<eat::bf16_t*>(out);
  EAT_DYRELU2_DISPATCH(dyrelu_ca_fwd2_kernel, eat::bf16_t, z16, a, b, coef, gate_f, gate_t, o16, n_planes, C, Fo, To);
  return eat::check_launch("eat_dyrelu_ca_fwd2_b16");
}

// backward: dv (w.r.t. v), dcoef (B,C,4), dgate_f / dgate_t (pre-sigmoid, layouts of the gates), bnpart (B,C,2) = per-plane
// (sum dv, sum dv * z) or NULL
extern "C" int eat_dyrelu_ca_bwd2(const float* dout, const float* z, const float* a, const float* b, const float* coef,
                                  const float* gate_f, const float* gate_t, float* dv, float* dcoef, float* dgate_f,
                                  float* dgate_t, float* bnpart, int B, int C, int Fo, int To, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || Fo < 1 || To < 1 || To > 512) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_bwd2: bad shape (To <= 512)");
  if ((a == nullptr) != (b == nullptr)) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_bwd2: a and b come together");
  hipStream_t hs = (hipStream_t)stream;
  EAT_DYRELU2_DISPATCH(dyrelu_ca_bwd2_kernel, float, dout, z, a, b, coef, gate_f, gate_t, dv, dcoef, dgate_f, dgate_t, bnpart, n_planes, C, Fo, To);
  return eat::check_launch("eat_dyrelu_ca_bwd2");
}

// bf16-storage twin: dout, z and dv are bf16 in HBM (bnpart: sums of dv as stored)
extern "C" int eat_dyrelu_ca_bwd2_b16(const void* dout, const void* z, const float* a, const float* b, const float* coef,
                                      const float* gate_f, const float* gate_t, void* dv, float* dcoef, float* dgate_f,
                                      float* dgate_t, float* bnpart, int B, int C, int Fo, int To, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || Fo < 1 || To < 1 || To > 512) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_bwd2_b16: bad shape (To <= 512)");
  if ((a == nullptr) != (b == nullptr)) return eat::fail(EAT_EINVAL, "eat_dyrelu_ca_bwd2_b16: a and b come together");
  hipStream_t hs = (hipStream_t)stream;
  const eat::bf16_t* d16 = reinterpret_cast<const eat::bf16_t*>(dout);
  const eat::bf16_t* z16 = reinterpret_cast<const eat::bf16_t*>(z);
  eat::bf16_t* dv16 = reinterpret_cast<eat::bf16_t*>(dv);
  EAT_DYRELU2_DISPATCH(dyrelu_ca_bwd2_kernel, eat::bf16_t, d16, z16, a, b, coef, gate_f, gate_t, dv16, dcoef, dgate_f, dgate_t, bnpart, n_planes, C, Fo, To);
  return eat::check_launch("eat_dyrelu_ca_bwd2_b16");
}
#undef EAT_DYRELU2_DISPATCH
#undef EAT_DYRELU2_LAUNCH

// Channel sums of a BatchNorm backward from per-plane partials (see dyn_bn_bwd_combine_kernel): p0 / p1 point at the first
// element of the two partial arrays, element (b, c, i) lies stride_e * ((b*C + c)*inner + i) floats further on.
extern "C" int eat_bn_bwd_combine_partials(const float* p0, const float* p1, int stride_e, int B, int C, int inner,
                                           const float* mean, const float* invstd, double* sums, float* dgamma, float* dbeta,
                                           eat_stream_t stream) {
  eat::clear_stale_error();
  if (!p0 || !p1 || stride_e < 1 || B < 1 || C < 1 || inner < 1) return eat::fail(EAT_EINVAL, "eat_bn_bwd_combine_partials: bad arguments");
  hipLaunchKernelGGL(dyn_bn_bwd_combine_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, p0, p1, stride_e, B, C, inner, mean,
                     invstd, sums, dgamma, dbeta);
  return eat::check_launch("eat_bn_bwd_combine_partials");
}

// ---- pointwise tails of the Linear layers that read the pooled context h_c (round 6) -----------------------------------------
// y (B, n_att*K + 4*cexp) = h_c [W_att_1; ...; W_att_n; W_coef]^T + bias (one GEMM, eat_linear_fwd):
//   att[i][b][:]   = softmax(y[b, i*K : (i+1)*K] / T_i)                       kernel attention of DynamicConv i (dy_block.py:106-109)
//   sg[b][j]       = sigmoid(y[b, n_att*K + j])                                kept for the backward
//   coef[b][c][m]  = (2 sg[b][4c+m] - 1) * lambdas[m] + init_v[m]              DyReLU-B coefficients (dy_block.py:176-181)
// Before: ~8 (forward) + ~12 (backward) elementwise / softmax / cat launches of ~5 us per block per step.
namespace {

__global__ __launch_bounds__(256) void dyn_heads_fwd_kernel(const float* __restrict__ y, int n_att, int K, int cexp, float it0,
                                                            float it1, float it2, const float* __restrict__ lambdas,
                                                            const float* __restrict__ init_v, float* __restrict__ att,
                                                            float* __restrict__ sg, float* __restrict__ coef, int B) {
  const int b = blockIdx.x, W = n_att * K + 4 * cexp;
  const float* yb = y + (size_t)b * W;
  if (threadIdx.x < n_att) {                     // one lane per attention head: K (<= 32) logits
    const int i = threadIdx.x;
    const float it = i == 0 ? it0 : (i == 1 ? it1 : it2);
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, yb[i * K + k] * it);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += expf(yb[i * K + k] * it - mx);
    const float inv = 1.0f / s;
    for (int k = 0; k < K; ++k) att[((size_t)i * B + b) * K + k] = expf(yb[i * K + k] * it - mx) * inv;
  }
  for (int j = threadIdx.x; j < 4 * cexp; j += blockDim.x) {
    const float s = 1.0f / (1.0f + expf(-yb[n_att * K + j]));
    sg[(size_t)b * 4 * cexp + j] = s;
    coef[(size_t)b * 4 * cexp + j] = fmaf(2.0f * s - 1.0f, lambdas[j & 3], init_v[j & 3]);
  }
}

__global__ __launch_bounds__(256) void dyn_heads_bwd_kernel(const float* __restrict__ datt, const float* __restrict__ dcoef,
                                                            const float* __restrict__ att, const float* __restrict__ sg,
                                                            const float* __restrict__ lambdas, int n_att, int K, int cexp,
                                                            float it0, float it1, float it2, float* __restrict__ dy, int B) {
  const int b = blockIdx.x, W = n_att * K + 4 * cexp;
  float* db = dy + (size_t)b * W;
  if (threadIdx.x < n_att) {
    const int i = threadIdx.x;
    const float it = i == 0 ? it0 : (i == 1 ? it1 : it2);
    const float* a = att + ((size_t)i * B + b) * K;
    const float* d = datt + ((size_t)i * B + b) * K;
    float dot = 0.0f;
    for (int k = 0; k < K; ++k) dot = fmaf(d[k], a[k], dot);
    for (int k = 0; k < K; ++k) db[i * K + k] = a[k] * (d[k] - dot) * it;
  }
  for (int j = threadIdx.x; j < 4 * cexp; j += blockDim.x) {
    const float s = sg[(size_t)b * 4 * cexp + j];
    db[n_att * K + j] = dcoef[(size_t)b * 4 * cexp + j] * lambdas[j & 3] * (2.0f * s * (1.0f - s));
  }
}

}  // namespace

extern "C" int eat_dyn_heads_fwd(const float* y, int B, int n_att, int K, int cexp, float inv_t0, float inv_t1, float inv_t2,
                                 const float* lambdas, const float* init_v, float* att, float* sg, float* coef,
                                 eat_stream_t stream) {
  eat::clear_stale_error();
  if (!y || !lambdas || !init_v || !att || !sg || !coef || B < 1 || n_att < 1 || n_att > 3 || K < 1 || K > 32 || cexp < 1)
    return eat::fail(EAT_EINVAL, "eat_dyn_heads_fwd: bad arguments (1 <= n_att <= 3, K <= 32)");
  hipLaunchKernelGGL(dyn_heads_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, y, n_att, K, cexp, inv_t0, inv_t1, inv_t2,
                     lambdas, init_v, att, sg, coef, B);
  return eat::check_launch("eat_dyn_heads_fwd");
}

extern "C" int eat_dyn_heads_bwd(const float* datt, const float* dcoef, const float* att, const float* sg, const float* lambdas,
                                 int B, int n_att, int K, int cexp, float inv_t0, float inv_t1, float inv_t2, float* dy,
                                 eat_stream_t stream) {
  eat::clear_stale_error();
  if (!datt || !dcoef || !att || !sg || !lambdas || !dy || B < 1 || n_att < 1 || n_att > 3 || K < 1 || K > 32 || cexp < 1)
    return eat::fail(EAT_EINVAL, "eat_dyn_heads_bwd: bad arguments (1 <= n_att <= 3, K <= 32)");
  hipLaunchKernelGGL(dyn_heads_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, datt, dcoef, att, sg, lambdas, n_att, K,
                     cexp, inv_t0, inv_t1, inv_t2, dy, B);
  return eat::check_launch("eat_dyn_heads_bwd");
}
