// Pointwise (1x1) convolution and Linear layers as fp32 MFMA GEMMs for gfx950.
//   reference: models/mn/block_types.py:138-147 (expand), :167-171 (project), :83 (SE scale),
//              :177-181 (residual); models/mn/model.py:159-167 (last conv), :186-194 (head)
//
// y (Co, N) = W (Co, Ci) . x (Ci, N) with N = B*S the flattened (sample, position) axis: NCHW keeps
// S = F*T contiguous per channel, so a k-row of a 256-column tile is (pieces of) contiguous
// memory.  The activations are the MFMA B operand, the weights the A operand, pre-packed once
// (BatchNorm scale folded in) into fragment order.
//
// Block = 4 waves = (MTW*16 rows) x 256 columns; wave w owns columns [64w, 64w+64).  K is walked
// in chunks of 32 through a 2-stage LDS ring filled by LDS-DMA (global_load_lds, 16 B per lane:
// one wave instruction moves one k-row of the tile = 1 KiB, coalesced along time); the chunk
// c+1 loads fly while chunk c is multiplied.  B fragments are read with ds_read_b128 (row stride
// 256 floats: conflict-free for the b128 lane groups), A fragments with ds_read_b32.
// v_mfma_f32_16x16x4_f32 is an exact fp32 fmaf chain - what the 1e-3 logit-parity budget assumes.
//
// Column permutation: lane l holds x[k][n0 + 4*(l&15) + j] in element j of its float4 and feeds
// it to MFMA n-tile j, so in the C/D layout lane l holds, for a fixed row, 4 consecutive columns:
// the epilogue (bias, activation, residual, pooled sums) loads/stores float4.
#include "eat_common.h"
#include "pw_epilogue.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kKC = 32;          // max k rows per LDS stage
constexpr int kTileN = 256;      // columns per block

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16 B per lane global -> LDS (dst = wave-uniform base + lane*16)
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

// Pack W (Co, Ci) [* row_scale] into wp[(ks * MT + mt) * 64 + lane] = W[mt*16 + (lane&15)][ks*4 + (lane>>4)]
// trans: w is stored (Ci, Co) and its transpose is packed (the data-gradient GEMM uses W^T: no separate transpose copy)
__global__ void pw_prepack_kernel(const float* __restrict__ w, const float* __restrict__ row_scale,
                                  float* __restrict__ wp, int Co, int Ci, int MT, int trans) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = (Ci / 4) * MT * 64;
  if (i >= total) return;
  const int lane = i & 63, mt = (i >> 6) % MT, ks = (i >> 6) / MT;
  const int m = mt * 16 + (lane & 15), k = ks * 4 + (lane >> 4);
  float v = 0.0f;
  if (m < Co) v = (trans ? w[(size_t)k * Co + m] : w[(size_t)m * Ci + k]) * (row_scale ? row_scale[m] : 1.0f);
  wp[i] = v;
}

// LDS stage = [A: (kc/4)*MTW fragment rows of 64 floats, padded to 256][X: kc rows x 256][scale: kc x NS]
__host__ __device__ inline int a_stage_floats(int mtw, int kc) { return (((kc >> 2) * mtw * 64) + 255) & ~255; }
__host__ __device__ inline int stage_floats(int mtw, int kc, int ns) {
  return a_stage_floats(mtw, kc) + kc * kTileN + ((kc * ns + 3) & ~3);
}

// The same LDS-DMA issued from inline asm.  hipcc tracks a builtin-issued LDS-DMA as a pending LDS
// write and puts "s_waitcnt vmcnt(0)" in front of the next ds_read of the same array, which drains
// the chunks that are supposed to stay in flight.  An asm-issued DMA is invisible to that pass; the
// kernel counts the VM queue by hand (wait_vmcnt) and orders LDS visibility with its own barrier.
__device__ __forceinline__ unsigned lds_addr_uniform(float* lds_wave_base) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)lds_wave_base);
}
__device__ __forceinline__ void glds16_raw(const float* g, float* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(lds_addr_uniform(lds_wave_base)), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void glds4_raw(const float* g, float* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
               :: "s"(lds_addr_uniform(lds_wave_base)), "v"(g) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PIPE = true : 16-row chunks, ring of up to 3 LDS stages, chunk c+2 is issued while chunk c is
//               multiplied; every wave issues a FIXED number of LDS-DMA instructions per chunk
//               (kPipeKC/4 x-rows + ceil(MTW/4) A pieces [+1 scale piece]) so the wait for chunk c
//               is a counted s_waitcnt that leaves chunk c+1 in flight across the raw s_barrier.
//               Requires 16*NS <= 64.
// PIPE = false: generic fallback (any kc, any NS): 2 stages, vmcnt(0) + __syncthreads per chunk.
constexpr int kPipeKC = 16;

// TF: the conv input is act_in(tf.a[k] * x + tf.b[k]) evaluated between the LDS read and the MFMA (training: BatchNorm +
// activation of the depthwise conv fused into the project conv; see conv_pw_bf16.hip)
struct PwTf { const float* a; const float* b; int act; };

template <int MTW, bool PIPE, bool TF>
__global__ __launch_bounds__(256, 2) void pw_conv_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ pool, int B, int Ci, int Co, int S, int MT, int MC, int n_tiles, int NS, int kc_arg,
    int n_stages, int act, int tps, long long wp_bstride, PwTf tf, const float* __restrict__ x2, int c1,
    float* __restrict__ stats, eat::PwGStat gs) {
  // gs.z != NULL (with stats): the partials are the BatchNorm-backward sums of pw_epilogue_gstats instead
  // x2 != NULL: channels of x (c1 rows) followed by the channels of x2 (Ci - c1 rows) - see conv_pw_bf16.hip
  // stats != NULL: per-tile partial sums of the output for the BatchNorm that follows (pw_epilogue_stats)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int kc = PIPE ? kPipeKC : kc_arg;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // XCD-aware mapping: the MC row-chunks that share one column tile sit on the same XCD (ids
  // congruent mod 8) so the tile's re-reads hit that XCD's L2.
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
  const int mchunk = jj % MC, tile = (jj / MC) * 8 + xcd;
  if (tile >= n_tiles) return;
  const int mt0 = mchunk * MTW;
  // tps == 0: the N axis is the flattened (sample, position) axis and tiles may straddle samples.
  // tps  > 0: per-sample weights (DyMN dynamic conv, models/dymn/dy_block.py:111-127): a tile lies
  //           inside one sample (tps tiles per sample) and reads that sample's packed weights.
  const long long n_base = tps ? (long long)(tile / tps) * S + (long long)(tile % tps) * kTileN : (long long)tile * kTileN;
  const int b_first = (int)(n_base / S);
  const long long N = tps ? (long long)(b_first + 1) * S : (long long)B * S;   // first column this tile must not touch
  wp += tps ? (size_t)b_first * wp_bstride : 0;

  // loader role: this lane's 4 columns of every k-row
  long long nl = n_base + 4 * lane;
  if (nl > N - 4) nl = N - 4;                       // clamp: garbage columns are never stored
  const int bl = (int)(nl / S), sl = (int)(nl - (long long)bl * S);
  const float* xsrc = x + ((size_t)bl * (x2 ? c1 : Ci)) * S + sl;
  const float* xsrc2 = x2 ? x2 + ((size_t)bl * (Ci - c1)) * S + sl : xsrc;
  auto xrow = [&](int row) { return (x2 && row >= c1) ? xsrc2 + (size_t)(row - c1) * S : xsrc + (size_t)row * S; };
  // compute role
  const long long nc = n_base + 64 * wv + 4 * (lane & 15);
  const bool col_ok = nc < N;
  const long long ncc = col_ok ? nc : N - 4;
  const int bc = (int)(ncc / S), sc_ = (int)(ncc - (long long)bc * S);
  const int kq = lane >> 4;

  const int a_sz = a_stage_floats(MTW, kc);
  const int stage_sz = PIPE ? a_sz + kc * kTileN + (in_scale ? 64 : 0) : stage_floats(MTW, kc, NS);
  const int n_chunks = (Ci + kc - 1) / kc;

  auto issue = [&](int c) {
    float* st = smem + (PIPE ? (c % n_stages) : (c & 1)) * stage_sz;
    const int k0 = c * kc;
    const int klen = (Ci - k0) < kc ? (Ci - k0) : kc;
    float* Xs = st + a_sz;
    const int rows = (klen >> 2) * MTW;     // A: fragment rows of 64 floats; 256 floats per wave instruction
    if constexpr (PIPE) {
#pragma unroll
      for (int i = 0; i < kPipeKC / 4; ++i) {
        const int r = wv + 4 * i;
        const int rc = r < klen ? r : klen - 1;          // tail chunk: re-load a valid row into an unused slot
        glds16_raw(xrow(k0 + rc), Xs + r * kTileN);
      }
#pragma unroll
      for (int i = 0; i < (MTW + 3) / 4; ++i) {
        int q = wv + 4 * i;
        if (q * 256 >= a_sz) q = 0;                        // keep the per-wave count fixed: re-load piece 0
        int row = q * 4 + (lane >> 4);
        if (row >= rows) row = rows - 1;
        const int ksl = row / MTW, ii = row - ksl * MTW;
        int mt = mt0 + ii;
        if (mt >= MT) mt = MT - 1;
        glds16_raw(wp + ((size_t)((k0 >> 2) + ksl) * MT + mt) * 64 + 4 * (lane & 15), st + q * 256);
      }
      if (in_scale) {
        int e = lane < kPipeKC * NS ? lane : kPipeKC * NS - 1;
        const int r = e / NS, j = e - r * NS;
        int bb = b_first + j;
        if (bb >= B) bb = B - 1;
        const int rc = r < klen ? r : klen - 1;
        glds4_raw(in_scale + (size_t)bb * Ci + k0 + rc, Xs + kc * kTileN);   // 64 floats reserved (NS <= 4)
      }
    } else {
      for (int r = wv; r < klen; r += 4) glds16(xrow(k0 + r), Xs + r * kTileN);
      const int n_inst = (rows + 3) >> 2;
      for (int q = wv; q < n_inst; q += 4) {
        int row = q * 4 + (lane >> 4);
        if (row >= rows) row = rows - 1;
        const int ksl = row / MTW, i = row - ksl * MTW;
        int mt = mt0 + i;
        if (mt >= MT) mt = MT - 1;
        glds16(wp + ((size_t)((k0 >> 2) + ksl) * MT + mt) * 64 + 4 * (lane & 15), st + q * 256);
      }
      if (in_scale) {
        float* SCs = Xs + kc * kTileN;
        for (int e = tid; e < klen * NS; e += 256) {
          const int r = e / NS, j = e - r * NS;
          int bb = b_first + j;
          if (bb >= B) bb = B - 1;
          SCs[e] = in_scale[(size_t)bb * Ci + k0 + r];
        }
      }
    }
  };

  f32x4 acc[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  __shared__ float s_bias[128];
  eat::pw_stage_bias(bias, s_bias, mt0, Co, wv, lane);    // older than every chunk: landed when chunk 0 is
  issue(0);
  if (PIPE && n_chunks > 1) issue(1);
  for (int c = 0; c < n_chunks; ++c) {
    if constexpr (PIPE) {
      constexpr int kLoads = kPipeKC / 4 + (MTW + 3) / 4;   // LDS-DMA instructions per wave per chunk
      if (c + 1 < n_chunks) {                                // leave chunk c+1 in flight
        if (in_scale) wait_vmcnt<kLoads + 1>(); else wait_vmcnt<kLoads>();
      } else {
        wait_vmcnt<0>();
      }
      // raw barrier: __syncthreads() would drain the LDS-DMA of chunk c+1 (vmcnt(0))
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (c + 2 < n_chunks) issue(c + 2);
    } else {
      wait_vmcnt<0>();
      __syncthreads();
      if (c + 1 < n_chunks) issue(c + 1);
    }
    const float* st = smem + (PIPE ? (c % n_stages) : (c & 1)) * stage_sz;
    const float* Xw = st + a_sz + 64 * wv + 4 * (lane & 15) + kq * kTileN;
    const float* SCs = st + a_sz + kc * kTileN + kq * NS + (bc - b_first);
    const float* Aw = st + lane;
    const int k0 = c * kc;
    const int ksteps = ((Ci - k0) < kc ? (Ci - k0) : kc) >> 2;
    // one software-pipelined loop: the fragments of k-step ks+1 are read from LDS while the
    // MFMAs of k-step ks issue
    float4 xv = *reinterpret_cast<const float4*>(Xw);
    float sv = in_scale ? SCs[0] : 1.0f;
    float tav = TF ? tf.a[k0 + kq] : 1.0f, tbv = TF ? tf.b[k0 + kq] : 0.0f;     // this lane's row k = k0 + 4 ks + kq
    const eat::ActCoef tac = eat::act_coef(TF ? tf.act : 0);
    float a[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) a[i] = Aw[i * 64];
    for (int ks = 0; ks < ksteps; ++ks) {
      const int kn = (ks + 1 < ksteps) ? ks + 1 : ks;
      const float4 xn = *reinterpret_cast<const float4*>(Xw + kn * 4 * kTileN);
      const float sn = in_scale ? SCs[kn * 4 * NS] : 1.0f;
      const float tan_ = TF ? tf.a[k0 + 4 * kn + kq] : 1.0f, tbn_ = TF ? tf.b[k0 + 4 * kn + kq] : 0.0f;
      float an[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) an[i] = Aw[(kn * MTW + i) * 64];
      if constexpr (TF) {
        xv.x = eat::act_apply(fmaf(tav, xv.x, tbv), tac); xv.y = eat::act_apply(fmaf(tav, xv.y, tbv), tac);
        xv.z = eat::act_apply(fmaf(tav, xv.z, tbv), tac); xv.w = eat::act_apply(fmaf(tav, xv.w, tbv), tac);
      }
      const float b0 = xv.x * sv, b1 = xv.y * sv, b2 = xv.z * sv, b3 = xv.w * sv;
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        acc[i][0] = mfma16(a[i], b0, acc[i][0]);
        acc[i][1] = mfma16(a[i], b1, acc[i][1]);
        acc[i][2] = mfma16(a[i], b2, acc[i][2]);
        acc[i][3] = mfma16(a[i], b3, acc[i][3]);
      }
      xv = xn; sv = sn; tav = tan_; tbv = tbn_;
#pragma unroll
      for (int i = 0; i < MTW; ++i) a[i] = an[i];
    }
  }

  if (stats && gs.z) {                                     // block-uniform
    // BEFORE the stores of y: loads and stores retire through one in-order counter - a z load issued behind the tile's
    // stores would wait for all of them (measured: the step 0.43 ms SLOWER than with the separate reduce pass)
    __syncthreads();                                       // every wave is done with the operand stages: LDS is free
    eat::pw_epilogue_gstats<MTW>(acc, s_bias, smem, stats, gs, tile, mt0, kq, lane, wv, col_ok, bc, sc_, Co, S);
  }
  eat::pw_epilogue<MTW>(acc, s_bias, res, y, pool, mt0, kq, lane, col_ok, bc, sc_, Co, S, act);
  if (stats && !gs.z) {
    __syncthreads();
    eat::pw_epilogue_stats<MTW>(acc, s_bias, smem, stats, tile, mt0, kq, lane, wv, col_ok, Co);
  }
}

// y (M,N) = act((x (M,K) * xs) . w (N,K)^T + bias): both operands K-contiguous; each lane loads 4
// consecutive k as one float4 and the 4 MFMAs of a 16-k step consume a consistent k permutation.
// Block = 4 ... 16 waves on ONE 16 x 32 output tile with K split over the waves (these GEMMs are tiny and
// latency-bound: M = batch, K,N <= a few thousand), partial sums combined through LDS in a fixed order.
// Round 4: up to 16 waves (one 64-k step each for K <= 1024) - with 4 waves the SE gate GEMMs (K = 960) ran four
// dependent load -> MFMA rounds per wave: 11.5 us per launch in the captured step, 50 launches per training step.
__global__ __launch_bounds__(1024) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int M, int K, int N, float xs, int act) {
  __shared__ float s_red[15][2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 32;
  const int row = lane & 15, kq = lane >> 4;
  const int mrow = m0 + row;
  const int kslice = (((K + nw - 1) / nw + 15) / 16) * 16; // per-wave K range, multiple of 16
  const int kb = wv * kslice, ke = (kb + kslice) < K ? (kb + kslice) : K;
  const bool vec = (K & 3) == 0;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  auto load4 = [&](const float* base, int r, int rmax, int k, float (&o)[4]) {
    o[0] = o[1] = o[2] = o[3] = 0.f;
    if (r >= rmax || k >= ke) return;
    if (vec) {
      const float4 t = *reinterpret_cast<const float4*>(base + (size_t)r * K + k);
      o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (k + e < ke) o[e] = base[(size_t)r * K + k + e];
    }
  };

  for (int k0 = kb; k0 < ke; k0 += 64) {
    float xa[4][4], wb[4][2][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {           // issue every load of 4 k-steps before the first MFMA
      const int k = k0 + 16 * u + 4 * kq;
      load4(x, mrow, M, k, xa[u]);
      load4(w, n0 + row, N, k, wb[u][0]);
      load4(w, n0 + 16 + row, N, k, wb[u][1]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = mfma16(xa[u][e] * xs, wb[u][0][e], acc[0]);
        acc[1] = mfma16(xa[u][e] * xs, wb[u][1][e], acc[1]);
      }
  }
  if (wv > 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_red[wv - 1][j][r][lane] = acc[j][r];
  }
  __syncthreads();
  if (wv != 0) return;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 16 * j + row;          // C/D: col = lane&15, row = kq*4 + r
    const float bn = (bias && n < N) ? bias[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + kq * 4 + r;
      float v = acc[j][r];
      for (int q = 0; q + 1 < nw; ++q) v += s_red[q][j][r][lane];
      v += bn;
      if (act == EAT_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
      else v = eat::activate_rt(v, act);
      if (n < N && m < M) y[(size_t)m * N + n] = v;
    }
  }
}

template <int MTW>
int launch_pw(hipStream_t s, const float* x, const float* wp, const float* bias, const float* in_scale,
              const float* res, float* y, float* pool, int B, int Ci, int Co, int S, int MT, int MC, int act,
              bool per_sample, PwTf tf, const float* x2, int c1, float* stats, eat::PwGStat gs) {
  const long long N = (long long)B * S;
  const int tps = per_sample ? (S + kTileN - 1) / kTileN : 0;
  const int n_tiles = per_sample ? B * tps : (int)((N + kTileN - 1) / kTileN);
  const long long wp_bstride = (long long)(Ci / 4) * MT * 64;
  int NS = kTileN / S + 2;
  if (NS > B) NS = B;
  if (!in_scale) NS = 0;
  const bool pipe = kPipeKC * NS <= 64;
  int kc, n_stages;
  size_t smem;
  if (pipe) {
    kc = kPipeKC;
    const int n_chunks = (Ci + kc - 1) / kc;
    n_stages = n_chunks < 3 ? n_chunks : 3;
    // the scale slot is a fixed 64-float piece in this mode
    smem = (size_t)n_stages * (a_stage_floats(MTW, kc) + kc * kTileN + (in_scale ? 64 : 0)) * sizeof(float);
  } else {
    // generic: <= 32 rows per stage, chunks balanced, multiple of 4
    const int n_chunks = (Ci + kKC - 1) / kKC;
    kc = (((Ci + n_chunks - 1) / n_chunks) + 3) & ~3;
    n_stages = ((Ci + kc - 1) / kc) > 1 ? 2 : 1;
    smem = (size_t)n_stages * stage_floats(MTW, kc, NS) * sizeof(float);
  }
  if (stats && smem < (size_t)10 * MTW * 16 * sizeof(float)) smem = (size_t)10 * MTW * 16 * sizeof(float);   // wave partials + (a, b)
  if (smem > 160 * 1024) return eat::fail(EAT_EINVAL, "eat_pw_conv_fwd: LDS stage too large (%zu B)", smem);
  auto kern = tf.a ? (pipe ? pw_conv_kernel<MTW, true, true> : pw_conv_kernel<MTW, false, true>)
                   : (pipe ? pw_conv_kernel<MTW, true, false> : pw_conv_kernel<MTW, false, false>);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eat::fail(EAT_ELAUNCH, "eat_pw_conv_fwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  const int tiles8 = (n_tiles + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(tiles8 * MC), dim3(256), smem, s, x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, MT,
                     MC, n_tiles, NS, kc, n_stages, act, tps, wp_bstride, tf, x2, c1, stats, gs);
  return eat::check_launch("eat_pw_conv_fwd");
}

}  // namespace

static int pw_prepack_impl(const float* w, const float* row_scale, float* wp, int Co, int Ci, int trans,
                           eat_stream_t stream) {
  eat::clear_stale_error();
  if (Ci % 4 != 0) return eat::fail(EAT_EINVAL, "eat_pw_prepack: Ci=%d must be a multiple of 4", Ci);
  const int MT = (Co + 15) / 16;
  const int total = (Ci / 4) * MT * 64;
  hipLaunchKernelGGL(pw_prepack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, row_scale,
                     wp, Co, Ci, MT, trans);
  return eat::check_launch("eat_pw_prepack");
}

extern "C" int eat_pw_prepack(const float* w, const float* row_scale, float* wp, int Co, int Ci,
                              eat_stream_t stream) {
  return pw_prepack_impl(w, row_scale, wp, Co, Ci, 0, stream);
}

extern "C" int eat_pw_prepack_t(const float* w_t, const float* row_scale, float* wp, int Co, int Ci,
                                eat_stream_t stream) {
  return pw_prepack_impl(w_t, row_scale, wp, Co, Ci, 1, stream);
}

static int pw_dispatch(const float* x, const float* wp, const float* bias, const float* in_scale, const float* res,
                       float* y, float* pool, int B, int Ci, int Co, int S, int act, bool per_sample, hipStream_t s,
                       PwTf tf = PwTf{nullptr, nullptr, 0}, const float* x2 = nullptr, int c1 = 0, float* stats = nullptr,
                       eat::PwGStat gs = eat::PwGStat{nullptr, nullptr, nullptr, 0}) {
  if (Ci % 4 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_fwd: Ci=%d must be a multiple of 4", Ci);
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_fwd: bad act %d", act);
  if (B < 1 || Ci < 4 || Co < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_pw_conv_fwd: bad shape");
  const int MT = (Co + 15) / 16;
  if (S % 4 != 0) {    // planes that do not start on 16-byte boundaries (e.g. 40-mel models): plain 4-byte kernel
    if (tf.a) return eat::fail(EAT_EINVAL, "eat_pw_conv_tf_fwd: S=%d must be a multiple of 4", S);
    if (stats) return 1;                           // no epilogue statistics there: the caller runs the separate pass
    return eat::pw_conv_generic(x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, act, 0,
                                per_sample ? (long long)(Ci / 4) * MT * 64 * (long long)sizeof(float) : 0, s);
  }
  // Row chunking.  Every block re-reads its 256-column x tile, and a CU takes in only ~10 B/clk, so
  // the tile must be tall: up to 8 m-tiles (128 rows) per block.
  const int MC = (MT + 7) / 8;                    // row chunks
  const int mtw = (MT + MC - 1) / MC;             // balanced m-tiles per block
#define EAT_PW_CASE(n) case n: return launch_pw<n>(s, x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, MT, (MT + n - 1) / n, act, per_sample, tf, x2, c1, stats, gs);
  switch (mtw) {
    EAT_PW_CASE(1) EAT_PW_CASE(2) EAT_PW_CASE(3) EAT_PW_CASE(4) EAT_PW_CASE(5)
    EAT_PW_CASE(6) EAT_PW_CASE(7) EAT_PW_CASE(8)
    default: return eat::fail(EAT_EINVAL, "eat_pw_conv_fwd: internal tiling error");
  }
#undef EAT_PW_CASE
}

extern "C" int eat_pw_conv_fwd(const float* x, const float* wp, const float* bias, const float* in_scale,
                               const float* res, float* y, float* pool, int B, int Ci, int Co, int S, int act,
                               eat_stream_t stream) {
  eat::clear_stale_error();
  return pw_dispatch(x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, act, false, (hipStream_t)stream);
}

// Train-mode project conv (models/mn/block_types.py:167-171 after :150-162): the conv input is
// act_in(tf_a[k] x + tf_b[k]) [* in_scale[b,k]] evaluated on load - BatchNorm + activation (+ SE scale) of the depthwise
// output are never materialised.  wmode: 0 = fp32 pack (eat_pw_prepack), 1 = bf16 pack, 2 = bf16 hi/lo pack.
extern "C" int eat_pw_conv_tf_fwd(const float* x, const float* tf_a, const float* tf_b, int tf_act, const void* wp,
                                  int wmode, const float* bias, const float* in_scale, const float* res, float* y, int B,
                                  int Ci, int Co, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!tf_a || !tf_b) return eat::fail(EAT_EINVAL, "eat_pw_conv_tf_fwd: tf_a and tf_b are required");
  if (tf_act < 0 || tf_act > 2 || act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_tf_fwd: bad activation code");
  if (Ci % 8 != 0 || S % 4 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_tf_fwd: needs Ci %% 8 == 0 and S %% 4 == 0 (Ci=%d, S=%d)", Ci, S);
  if (B < 1 || Co < 1) return eat::fail(EAT_EINVAL, "eat_pw_conv_tf_fwd: bad shape");
  if (wmode == 0)
    return pw_dispatch(x, reinterpret_cast<const float*>(wp), bias, in_scale, res, y, nullptr, B, Ci, Co, S, act, false,
                       (hipStream_t)stream, PwTf{tf_a, tf_b, tf_act});
  return eat::pw_conv_bf16_tf(x, tf_a, tf_b, tf_act, wp, bias, in_scale, res, y, B, Ci, Co, S, act, wmode == 2 ? 1 : 0,
                              (hipStream_t)stream);
}

// 1x1 conv over the channels of TWO tensors of the same (B, *, S) geometry: y = W [x1 ; x2] + bias (+ res), W = (Co,
// C1 + C2) packed as usual.  Train plan: the data gradient of the expand conv with its BatchNorm correction,
// dx = (diag(a) W)^T g + M x + c0 (eat_expand_bwd_coef), as ONE GEMM over [g ; x].  C1 % 4 == 0, C2 % 4 == 0, S % 4 == 0.
extern "C" int eat_pw_conv_cat_fwd(const float* x1, int C1, const float* x2, int C2, const void* wp, int wmode,
                                   const float* bias, const float* res, float* y, int B, int Co, int S, int act,
                                   eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x1 || !x2 || C1 < 4 || C2 < 4 || C1 % 4 || C2 % 4 || S % 4)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_cat_fwd: needs C1, C2, S multiples of 4 (C1=%d, C2=%d, S=%d)", C1, C2, S);
  if (act < 0 || act > 2 || B < 1 || Co < 1) return eat::fail(EAT_EINVAL, "eat_pw_conv_cat_fwd: bad arguments");
  if (wmode == 0)
    return pw_dispatch(x1, reinterpret_cast<const float*>(wp), bias, nullptr, res, y, nullptr, B, C1 + C2, Co, S, act, false,
                       (hipStream_t)stream, PwTf{nullptr, nullptr, 0}, x2, C1);
  return eat::pw_conv_bf16_cat(x1, C1, x2, C2, wp, bias, res, y, B, Co, S, act, wmode == 2 ? 1 : 0, (hipStream_t)stream);
}

extern "C" int eat_pw_conv_dyn_fwd(const float* x, const float* wp_b, const float* bias, const float* res, float* y,
                                   int B, int Ci, int Co, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  return pw_dispatch(x, wp_b, bias, nullptr, res, y, nullptr, B, Ci, Co, S, act, true, (hipStream_t)stream);
}

// Train-mode 1x1 conv (no bias / activation / residual: z = W x) with the partial sums of z for the BatchNorm that follows
// in its epilogue - see pw_epilogue_stats.  wmode: 0 = fp32 pack, 1 = bf16, 2 = bf16 hi / lo; per_sample: wp holds one
// pack per sample (eat_dyn_pw_pack / eat_dyn_pw_pack_bf16); tf_a / tf_b / tf_act, in_scale: as eat_pw_conv_tf_fwd (NULL:
// none).  part: eat_pw_conv_stat_tiles(B, S, per_sample) * 2 * Co floats, layout [tile][2][Co] = eat_bn_finalize_partials
// with outer = tiles, inner = 1.  Returns 1 (and launches nothing) for geometries without the epilogue (S % 4 != 0).
extern "C" int eat_pw_conv_stat_tiles(int B, int S, int per_sample) {
  return per_sample ? B * ((S + kTileN - 1) / kTileN) : (int)(((long long)B * S + kTileN - 1) / kTileN);
}

extern "C" int eat_pw_conv_stats_fwd(const float* x, const void* wp, int wmode, int per_sample, const float* tf_a,
                                     const float* tf_b, int tf_act, const float* in_scale, const float* zero_bias, float* y,
                                     float* part, int B, int Ci, int Co, int S, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !wp || !y || !part || !zero_bias) return eat::fail(EAT_EINVAL, "eat_pw_conv_stats_fwd: missing operand");
  if ((tf_a == nullptr) != (tf_b == nullptr) || tf_act < 0 || tf_act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_stats_fwd: bad transform");
  if (tf_a && Ci % 8 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_stats_fwd: the on-load transform needs Ci %% 8 == 0");
  if (S % 4 != 0) return 1;
  if (wmode == 0)
    return pw_dispatch(x, reinterpret_cast<const float*>(wp), zero_bias, in_scale, nullptr, y, nullptr, B, Ci, Co, S, EAT_ACT_NONE,
                       per_sample != 0, (hipStream_t)stream, PwTf{tf_a, tf_b, tf_act}, nullptr, 0, part);
  return eat::pw_conv_bf16_stats(x, wp, wmode == 2 ? 1 : 0, per_sample, tf_a, tf_b, tf_act, in_scale, zero_bias, y, part, B, Ci,
                                 Co, S, (hipStream_t)stream);
}

// Data-gradient GEMM of a project conv (y = dxs = W^T dz_p; wp = the transposed pack, wmode as above, no per-sample form)
// with the BatchNorm + activation BACKWARD statistics of the depthwise output z_d in its epilogue (pw_epilogue_gstats): part
// [tile][2][Co] partials of sum g and sum g z_d, g = dxs * act'(g_a z_d + g_b) - the reduce pass over (dxs, z_d) disappears;
// eat_bn_bwd_sums_from_tiles finishes.  Returns 1 (nothing launched) for S % 4 != 0.
extern "C" int eat_pw_conv_gstats_fwd(const float* x, const void* wp, int wmode, const float* zero_bias, float* y,
                                      const float* gz, const float* g_a, const float* g_b, int g_act, float* part, int B,
                                      int Ci, int Co, int S, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !wp || !y || !part || !zero_bias || !gz || !g_a || !g_b)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_gstats_fwd: missing operand");
  if (g_act < 0 || g_act > 2 || (wmode != 0 && wmode != 2)) return eat::fail(EAT_EINVAL, "eat_pw_conv_gstats_fwd: bad act / wmode");
  if (S % 4 != 0) return 1;
  const eat::PwGStat gs{gz, g_a, g_b, g_act};
  if (wmode == 0)
    return pw_dispatch(x, reinterpret_cast<const float*>(wp), zero_bias, nullptr, nullptr, y, nullptr, B, Ci, Co, S, EAT_ACT_NONE,
                       false, (hipStream_t)stream, PwTf{nullptr, nullptr, 0}, nullptr, 0, part, gs);
  return eat::pw_conv_bf16_stats(x, wp, 1, 0, nullptr, nullptr, 0, nullptr, zero_bias, y, part, B, Ci, Co, S, (hipStream_t)stream, gs);
}

extern "C" int eat_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                              float x_scale, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (act < 0 || act > 3) return eat::fail(EAT_EINVAL, "eat_linear_fwd: bad act %d", act);
  dim3 grid((N + 31) / 32, (B + 15) / 16);
  int nw = (K + 63) / 64;                                  // one 64-k step per wave where K allows
  nw = nw < 4 ? 4 : (nw > 16 ? 16 : nw);
  hipLaunchKernelGGL(linear_kernel, grid, dim3(64 * nw), 0, (hipStream_t)stream, x, w, bias, y, B, K, N, x_scale, act);
  return eat::check_launch("eat_linear_fwd");
}
