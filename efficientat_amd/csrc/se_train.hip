// Backward of the squeeze-excitation gate MLP in two launches (round 3).
//   reference: autograd through models/mn/block_types.py:72-83 - scale = sigmoid(fc2(relu(fc1(mean_{f,t} y)))), out = y * scale.
//
// Given ds[b,c] = sum_{f,t} d_out * y (the gate's incoming gradient, taken by eat_se_bn_bwd_partials / eat_plane_dot):
//   dq  = ds * s * (1 - s)                                   (B, C)
//   dW2 = dq^T h,  db2 = sum_b dq                            (C, Cr), (C)
//   dh  = (dq W2) * [h > 0]                                  (B, Cr)
//   dW1 = dh^T zmean,  db1 = sum_b dh,  zmean = pool / S     (Cr, C), (Cr)
//   gadd = (dh W1) / S                                       (B, C)   - the gradient w.r.t. y through the squeeze
// As torch ops this was ~25 launches per block (transposes for the contraction-contiguous GEMM kernel, elementwise
// products, sums, masks) of 3-6 us each on KB-sized tensors: 8 SE blocks = ~0.9 ms of a 27 ms step spent on launch
// latency.  Here: stage 1 = {dW2, db2, dh}, stage 2 = {dW1, db1, gadd}, each ONE launch whose blocks pick their problem
// from blockIdx.x; plain fp32 FMA on 32 x 32 tiles staged through LDS (the GEMMs are <= 120 MFLOP; operands stay in L2),
// operands addressed by strides so that no transposed copy is ever made.  Every output element is produced by one block
// (full contraction): no atomics, bit-reproducible.
#include "eat_common.h"

namespace {

// one 32 x 32 output tile: out(m, n) = sum_k A(m, k) B(k, n); A_MFAST: A is contiguous along m (else along k); B is
// contiguous along n.  256 threads = 4 waves, each one 16 x 16 quadrant on the fp32 matrix cores (v_mfma_f32_16x16x4_f32:
// fp32 products and accumulation - round 5; the round-3 form multiplied on the vector ALUs, 4 outputs per thread, which at
// mn40's widths - 3840 x 960 gates, a 3840 -> 5120 -> 527 head - ran at 14 TFLOP/s: 2.1 + 0.65 ms of a 46 ms step).  The k
// axis goes through LDS in chunks of 64 (one LDS stage of 2 x 64 x 33 floats SHARED by the tile functions of a kernel:
// 16.9 KB per block, 8 waves per SIMD.  History, same-box microbenchmarks at mn40's head 3840 -> 5120 -> 527, B = 128:
// chunks of 128 with one pair of arrays PER INSTANTIATION of this template - 68.6 KB, two blocks per CU - 442 us; one shared
// stage 257 us; chunks of 64 236 us.  Chunks of 32 at two blocks per CU had left the kernel waiting on one memory latency
// per 32 k: 55 us for K = 960.)
constexpr int kKC = 64;
// [k_lo, K): the slice of the contraction this block sums (split-K: the caller combines the slices in a fixed order)
// ONE LDS stage for every instantiation of gemm_tile / col_sum_tile inside a kernel (a block runs exactly one of them): as
// function-local statics each instantiation had its own pair of arrays - 68.6 KB per block, two blocks per CU.
typedef float tile_lds_t[kKC][33];
__device__ __forceinline__ tile_lds_t* tile_lds() {
  __shared__ float s_tiles[2][kKC][33];
  return s_tiles;
}
template <bool A_MFAST, class FA, class FB, class FS>
__device__ __forceinline__ void gemm_tile(int M, int N, int K, int m0, int n0, FA a_at, FB b_at, FS store, int k_lo = 0) {
  tile_lds_t* const lds = tile_lds();
  tile_lds_t& sA = lds[0];                           // sA[k][m]
  tile_lds_t& sB = lds[1];                           // sB[k][n]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int ms = 16 * (wv >> 1), ns = 16 * (wv & 1);                 // this wave's quadrant of the tile
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  for (int k0 = k_lo; k0 < K; k0 += kKC) {
    constexpr int NI = kKC / 8, KQ = kKC / 32;          // loads per thread and operand; 32-wide k groups of a chunk
    float av[NI], bv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      // A: fast index tx, slow index ty + 8 i over (m: 32) x (k: kKC)
      int mm, kk;
      if (A_MFAST) { mm = tx; kk = ty + 8 * i; }
      else { kk = tx + 32 * (i % KQ); mm = ty + 8 * (i / KQ); }
      av[i] = (m0 + mm < M && k0 + kk < K) ? a_at(m0 + mm, k0 + kk) : 0.0f;
      const int kb = ty + 8 * i;
      bv[i] = (k0 + kb < K && n0 + tx < N) ? b_at(k0 + kb, n0 + tx) : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int mm, kk;
      if (A_MFAST) { mm = tx; kk = ty + 8 * i; }
      else { kk = tx + 32 * (i % KQ); mm = ty + 8 * (i / KQ); }
      sA[kk][mm] = av[i];
      sB[ty + 8 * i][tx] = bv[i];
    }
    __syncthreads();
    // rows of the chunk beyond K were stored as zeros: whole 4-k steps
    const int kn = (K - k0) < kKC ? ((K - k0 + 3) & ~3) : kKC;
    // A operand: lane (m = l15, k = kq); B operand: lane (k = kq, n = l15); D: lane holds rows 4 kq + r of column l15
#pragma unroll 4
    for (int kk = 0; kk < kn; kk += 4)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sA[kk + kq][ms + l15], sB[kk + kq][ns + l15], acc, 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + ms + 4 * kq + r, n = n0 + ns + l15;
    if (m < M && n < N) store(m, n, acc[r]);
  }
}

// column sums out[n] = sum_m f(m, n) for 32 columns per block: 8 row groups x 32 columns, LDS reduction
template <class F>
__device__ __forceinline__ void col_sum_tile(int M, int N, int n0, F f, float* __restrict__ out) {
  tile_lds_t& s_cs = tile_lds()[0];                  // (8 rows of it)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float s = 0.0f;
  if (n0 + tx < N)
    for (int m = ty; m < M; m += 8) s += f(m, n0 + tx);
  s_cs[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n0 + tx < N) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += s_cs[i][tx];
    out[n0 + tx] = t;
  }
}

__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// Split-K for the GEMMs with a long contraction and few output tiles (round 5: dh = dq W2 at mn40's widths is 120 tiles of
// K = 3840 - 30 dependent load -> LDS -> MFMA rounds per block, 176 us; the head's dfeat = du W1 has K = 5120): a block sums
// one slice of k, the slices are stored side by side ([slice][M][N]) and added in index order by slice_sum_kernel - no
// atomics, bit-reproducible.  Slices of 256 from K = 512 (same-box: slices of 512 from K = 1024 left mn10's 960 -> 240 gate at
// 55 us, now 33 us; mn40's 3840 -> 960 gate 120 -> 112 us); one slice (no extra launch) below.
__host__ __device__ __forceinline__ int k_slices(int K) { return K >= 512 ? (K + 255) / 256 : 1; }
__host__ __device__ __forceinline__ int k_slice_len(int K) { const int n = k_slices(K); return ((K + n - 1) / n + 3) & ~3; }

// out[i] = (gate == NULL || gate[i] > 0) * sum over slices of part[s][i]
__global__ __launch_bounds__(256) void slice_sum_kernel(const float* __restrict__ part, const float* __restrict__ gate,
                                                        float* __restrict__ out, int n, int n_slices) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float t = 0.0f;
  for (int s = 0; s < n_slices; ++s) t += part[(size_t)s * n + i];
  out[i] = (gate == nullptr || gate[i] > 0.0f) ? t : 0.0f;
}

// stage 1: blocks [0, nA) -> dW2 tiles, [nA, nA + nB) -> dh tiles, then cdiv(C, 32) blocks -> db2
__global__ __launch_bounds__(256) void se_bwd_stage1_kernel(const float* __restrict__ ds, const float* __restrict__ sc,
                                                            const float* __restrict__ h, const float* __restrict__ W2,
                                                            float* __restrict__ dW2, float* __restrict__ db2,
                                                            float* __restrict__ dh, int B, int C, int Cr) {
  auto dq = [&](int b, int c) { const float s = sc[(size_t)b * C + c]; return ds[(size_t)b * C + c] * s * (1.0f - s); };
  const int tA_n = cdiv(Cr, 32), nA = cdiv(C, 32) * tA_n;
  const int ks = k_slices(C), kl = k_slice_len(C);
  const int tB_n = cdiv(Cr, 32), nB1 = cdiv(B, 32) * tB_n, nB = nB1 * ks;
  int id = blockIdx.x;
  if (id < nA) {                      // dW2 (C x Cr): A(m = c, k = b) = dq[b, c], B(k = b, n = r) = h[b, r]
    gemm_tile<true>(C, Cr, B, (id / tA_n) * 32, (id % tA_n) * 32, [&](int m, int k) { return dq(k, m); },
                    [&](int k, int n) { return h[(size_t)k * Cr + n]; },
                    [&](int m, int n, float v) { dW2[(size_t)m * Cr + n] = v; });
    return;
  }
  id -= nA;
  if (id < nB) {                      // dh (B x Cr): A(m = b, k = c) = dq[b, c], B(k = c, n = r) = W2[c, r]
    const int sl = id / nB1, t = id - sl * nB1;
    const int k_hi = (sl + 1) * kl < C ? (sl + 1) * kl : C;
    // one slice: the finished dh; several: slice sl of dh_part (= dh's buffer, [ks][B][Cr]; ReLU mask in slice_sum_kernel)
    float* o = dh + (size_t)sl * B * Cr;
    gemm_tile<false>(B, Cr, k_hi, (t / tB_n) * 32, (t % tB_n) * 32, [&](int m, int k) { return dq(m, k); },
                     [&](int k, int n) { return W2[(size_t)k * Cr + n]; },
                     [&](int m, int n, float v) { o[(size_t)m * Cr + n] = (ks > 1 || h[(size_t)m * Cr + n] > 0.0f) ? v : 0.0f; },
                     sl * kl);
    return;
  }
  id -= nB;
  col_sum_tile(B, C, id * 32, dq, db2);
}

// stage 2: blocks [0, nC) -> dW1 tiles, [nC, nC + nD) -> gadd tiles, then cdiv(Cr, 32) blocks -> db1
__global__ __launch_bounds__(256) void se_bwd_stage2_kernel(const float* __restrict__ dh, const float* __restrict__ pool,
                                                            const float* __restrict__ W1, float inv_s,
                                                            float* __restrict__ dW1, float* __restrict__ db1,
                                                            float* __restrict__ gadd, int B, int C, int Cr) {
  const int tC_n = cdiv(C, 32), nC = cdiv(Cr, 32) * tC_n;
  const int tD_n = cdiv(C, 32), nD = cdiv(B, 32) * tD_n;
  int id = blockIdx.x;
  if (id < nC) {                      // dW1 (Cr x C): A(m = r, k = b) = dh[b, r], B(k = b, n = c) = pool[b, c] / S
    gemm_tile<true>(Cr, C, B, (id / tC_n) * 32, (id % tC_n) * 32, [&](int m, int k) { return dh[(size_t)k * Cr + m]; },
                    [&](int k, int n) { return pool[(size_t)k * C + n] * inv_s; },
                    [&](int m, int n, float v) { dW1[(size_t)m * C + n] = v; });
    return;
  }
  id -= nC;
  if (id < nD) {                      // gadd (B x C): A(m = b, k = r) = dh[b, r], B(k = r, n = c) = W1[r, c]
    gemm_tile<false>(B, C, Cr, (id / tD_n) * 32, (id % tD_n) * 32, [&](int m, int k) { return dh[(size_t)m * Cr + k]; },
                     [&](int k, int n) { return W1[(size_t)k * C + n]; },
                     [&](int m, int n, float v) { gadd[(size_t)m * C + n] = v * inv_s; });
    return;
  }
  id -= nD;
  col_sum_tile(B, Cr, id * 32, [&](int b, int r) { return dh[(size_t)b * Cr + r]; }, db1);
}

// ---- the classifier head's backward on the same two-stage scheme (models/mn/model.py:186-194: Linear(C -> H), Hardswish,
// Dropout, Linear(H -> N)); round 5: replaces four rocBLAS GEMMs + five torch ops of the timed step.
//   stage 1: dW2 = dlogits^T h2 (N x H), db2 = sum_b dlogits, du = (dlogits W2) * mask * hardswish'(u)   (B x H)
//   stage 2: dW1 = du^T feat (H x C),   db1 = sum_b du,      dfeat = du W1                              (B x C)
__global__ __launch_bounds__(256) void head_bwd_stage1_kernel(const float* __restrict__ dl, const float* __restrict__ h2,
                                                              const float* __restrict__ u, const float* __restrict__ mask,
                                                              const float* __restrict__ W2, float* __restrict__ dW2,
                                                              float* __restrict__ db2, float* __restrict__ du, int B, int H,
                                                              int N) {
  const int tA_n = cdiv(H, 32), nA = cdiv(N, 32) * tA_n;
  const int tB_n = cdiv(H, 32), nB = cdiv(B, 32) * tB_n;
  int id = blockIdx.x;
  if (id < nA) {                      // dW2 (N x H): A(m = n, k = b) = dl[b, n], B(k = b, n = r) = h2[b, r]
    gemm_tile<true>(N, H, B, (id / tA_n) * 32, (id % tA_n) * 32, [&](int m, int k) { return dl[(size_t)k * N + m]; },
                    [&](int k, int n) { return h2[(size_t)k * H + n]; },
                    [&](int m, int n, float v) { dW2[(size_t)m * H + n] = v; });
    return;
  }
  id -= nA;
  if (id < nB) {                      // du (B x H): A(m = b, k = n) = dl[b, n], B(k = n, n = r) = W2[n, r]
    gemm_tile<false>(B, H, N, (id / tB_n) * 32, (id % tB_n) * 32, [&](int m, int k) { return dl[(size_t)m * N + k]; },
                     [&](int k, int n) { return W2[(size_t)k * H + n]; },
                     [&](int m, int n, float v) {
                       const float uv = u[(size_t)m * H + n];
                       // nn.Hardswish backward (PyTorch): 0 below -3, x / 3 + 1 / 2 inside [-3, 3], 1 above
                       const float d = uv < -3.0f ? 0.0f : (uv <= 3.0f ? uv * (1.0f / 3.0f) + 0.5f : 1.0f);
                       du[(size_t)m * H + n] = v * (mask ? mask[(size_t)m * H + n] : 1.0f) * d;
                     });
    return;
  }
  id -= nB;
  col_sum_tile(B, N, id * 32, [&](int b, int n) { return dl[(size_t)b * N + n]; }, db2);
}

__global__ __launch_bounds__(256) void head_bwd_stage2_kernel(const float* __restrict__ du, const float* __restrict__ feat,
                                                              const float* __restrict__ W1, float* __restrict__ dW1,
                                                              float* __restrict__ db1, float* __restrict__ dfeat, int B,
                                                              int C, int H) {
  const int tC_n = cdiv(C, 32), nC = cdiv(H, 32) * tC_n;
  const int ks = k_slices(H), kl = k_slice_len(H);
  const int tD_n = cdiv(C, 32), nD1 = cdiv(B, 32) * tD_n, nD = nD1 * ks;
  int id = blockIdx.x;
  if (id < nC) {                      // dW1 (H x C): A(m = r, k = b) = du[b, r], B(k = b, n = c) = feat[b, c]
    gemm_tile<true>(H, C, B, (id / tC_n) * 32, (id % tC_n) * 32, [&](int m, int k) { return du[(size_t)k * H + m]; },
                    [&](int k, int n) { return feat[(size_t)k * C + n]; },
                    [&](int m, int n, float v) { dW1[(size_t)m * C + n] = v; });
    return;
  }
  id -= nC;
  if (id < nD) {                      // dfeat (B x C): A(m = b, k = r) = du[b, r], B(k = r, n = c) = W1[r, c]
    const int sl = id / nD1, t = id - sl * nD1;
    const int k_hi = (sl + 1) * kl < H ? (sl + 1) * kl : H;
    float* o = dfeat + (size_t)sl * B * C;                      // (several slices: dfeat is the [ks][B][C] scratch)
    gemm_tile<false>(B, C, k_hi, (t / tD_n) * 32, (t % tD_n) * 32, [&](int m, int k) { return du[(size_t)m * H + k]; },
                     [&](int k, int n) { return W1[(size_t)k * C + n]; },
                     [&](int m, int n, float v) { o[(size_t)m * C + n] = v; }, sl * kl);
    return;
  }
  id -= nD;
  col_sum_tile(B, H, id * 32, [&](int b, int r) { return du[(size_t)b * H + r]; }, db1);
}

}  // namespace

static int cdiv_host(int a, int b) { return (a + b - 1) / b; }

// scratch sizes (floats) of the dh / dfeat buffers above: the result plus, when the contraction is split, its k slices
extern "C" int eat_se_mlp_dh_floats(int B, int C, int Cr) { return B * Cr * (k_slices(C) > 1 ? 1 + k_slices(C) : 1); }
extern "C" int eat_mlp_head_dfeat_floats(int B, int C, int H) { return B * C * (k_slices(H) > 1 ? 1 + k_slices(H) : 1); }

extern "C" int eat_mlp_head_bwd(const float* dlogits, const float* h2, const float* u, const float* drop_mask,
                                const float* feat, const float* W1, const float* W2, float* dW1, float* db1, float* dW2,
                                float* db2, float* du, float* dfeat, int B, int C, int H, int N, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!dlogits || !h2 || !u || !feat || !W1 || !W2 || !dW1 || !db1 || !dW2 || !db2 || !du || !dfeat)
    return eat::fail(EAT_EINVAL, "eat_mlp_head_bwd: missing operand");
  if (B < 1 || C < 1 || H < 1 || N < 1) return eat::fail(EAT_EINVAL, "eat_mlp_head_bwd: bad shape");
  // dfeat: eat_mlp_head_dfeat_floats(B, C, H) floats - the finished dfeat first, behind it the k slices (contraction over H)
  const int ks = k_slices(H);
  float* df_part = ks > 1 ? dfeat + (size_t)B * C : dfeat;
  const int n1 = cdiv_host(N, 32) * cdiv_host(H, 32) + cdiv_host(B, 32) * cdiv_host(H, 32) + cdiv_host(N, 32);
  const int n2 = cdiv_host(H, 32) * cdiv_host(C, 32) + cdiv_host(B, 32) * cdiv_host(C, 32) * ks + cdiv_host(H, 32);
  hipLaunchKernelGGL(head_bwd_stage1_kernel, dim3((unsigned)n1), dim3(256), 0, (hipStream_t)stream, dlogits, h2, u, drop_mask,
                     W2, dW2, db2, du, B, H, N);
  hipLaunchKernelGGL(head_bwd_stage2_kernel, dim3((unsigned)n2), dim3(256), 0, (hipStream_t)stream, du, feat, W1, dW1, db1,
                     df_part, B, C, H);
  if (ks > 1)
    hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, df_part,
                       (const float*)nullptr, dfeat, B * C, ks);
  return eat::check_launch("eat_mlp_head_bwd");
}

extern "C" int eat_se_mlp_bwd(const float* ds, const float* scale, const float* h, const float* pool, const float* W1,
                              const float* W2, float inv_s, float* dW1, float* db1, float* dW2, float* db2, float* dh,
                              float* gadd, int B, int C, int Cr, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1 || Cr < 1) return eat::fail(EAT_EINVAL, "eat_se_mlp_bwd: bad shape");
  // dh: scratch of eat_se_mlp_dh_floats(B, C, Cr) floats - the finished dh first, behind it the k slices when the
  // contraction over C is split (k_slices)
  const int ks = k_slices(C);
  float* dh_part = ks > 1 ? dh + (size_t)B * Cr : dh;
  const int n1 = cdiv_host(C, 32) * cdiv_host(Cr, 32) + cdiv_host(B, 32) * cdiv_host(Cr, 32) * ks + cdiv_host(C, 32);
  const int n2 = cdiv_host(Cr, 32) * cdiv_host(C, 32) + cdiv_host(B, 32) * cdiv_host(C, 32) + cdiv_host(Cr, 32);
  hipLaunchKernelGGL(se_bwd_stage1_kernel, dim3((unsigned)n1), dim3(256), 0, (hipStream_t)stream, ds, scale, h, W2, dW2, db2,
                     dh_part, B, C, Cr);
  if (ks > 1)
    hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((B * Cr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh_part, h, dh,
                       B * Cr, ks);
  hipLaunchKernelGGL(se_bwd_stage2_kernel, dim3((unsigned)n2), dim3(256), 0, (hipStream_t)stream, dh, pool, W1, inv_s, dW1,
                     db1, gadd, B, C, Cr);
  return eat::check_launch("eat_se_mlp_bwd");
}
