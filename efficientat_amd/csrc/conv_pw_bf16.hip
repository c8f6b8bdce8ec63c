// Pointwise (1x1) convolution on the bf16 matrix cores of gfx950 with fp32 activations in HBM.
//   v_mfma_f32_16x16x32_bf16 runs at 16x the fp32-input MFMA rate.  Two uses:
//     NPROD = 3  "bf16x3": x = x_hi + x_lo, w = w_hi + w_lo (each a bf16, round-to-nearest), and
//                y = w_hi x_hi + w_hi x_lo + w_lo x_hi accumulated in fp32: ~2^-16 relative error per
//                product at 3/16 of the fp32 MFMA time - for the compute-bound layers of the fp32 model;
//     NPROD = 1  plain bf16 operands, fp32 accumulation (BASELINE config 3: bf16 compute).
//   Same call sites as conv_pw.hip (models/mn/block_types.py:138-147,167-171,83,177-181).
//
// Structure = conv_pw.hip's: block = 4 waves = (MTW*16 rows) x 256 flattened (b,s) columns, K walked
// in 32-row chunks through a 2-stage LDS ring filled by LDS-DMA issued from inline asm (so hipcc does
// not drain the chunk in flight before every ds_read).  Per chunk a lane reads 8 rows x float4 of its
// 4 columns (ds_read_b128, conflict-free), splits them into bf16 hi/lo with v_cvt_pk_bf16_f32 and
// feeds 4 n-tiles; A fragments (hi/lo) were packed once on the device (eat_pw_prepack_bf16) so that a
// fragment is one 16-byte LDS read.  The fp32 epilogue (bias, activation, residual, pooled sums) is
// the one of conv_pw.hip.
#include <cstdlib>
#include <type_traits>
#include "eat_common.h"
#include "pw_epilogue.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kKC = 32;
constexpr int kTileN = 256;

__device__ __forceinline__ unsigned lds_addr_uniform(void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)p);
}
__device__ __forceinline__ void glds16_raw(const void* g, void* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(lds_addr_uniform(lds_wave_base)), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void glds4_raw(const void* g, void* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
               :: "s"(lds_addr_uniform(lds_wave_base)), "v"(g) : "memory", "m0");
}

// wp16[((kk*MT + mt)*NP2 + h)*512 + lane*8 + i] = bf16 part h of W[mt*16 + (lane&15)][kk*32 + 8*(lane>>4) + i]
__global__ void pw_prepack_bf16_kernel(const float* __restrict__ w, const float* __restrict__ row_scale,
                                       __bf16* __restrict__ wp, int Co, int Ci, int MT, int NP2, int trans) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (kk, mt, lane)
  const int KK = (Ci + 31) / 32;
  if (t >= KK * MT * 64) return;
  const int lane = t & 63, mt = (t >> 6) % MT, kk = (t >> 6) / MT;
  const int m = mt * 16 + (lane & 15), kb = kk * 32 + 8 * (lane >> 4);
  const float rs = (row_scale && m < Co) ? row_scale[m] : 1.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = kb + i;
    const float v = (m < Co && k < Ci) ? (trans ? w[(size_t)k * Co + m] : w[(size_t)m * Ci + k]) * rs : 0.0f;
    const __bf16 hi = (__bf16)v;
    wp[((size_t)(kk * MT + mt) * NP2 + 0) * 512 + lane * 8 + i] = hi;
    if (NP2 == 2) wp[((size_t)(kk * MT + mt) * NP2 + 1) * 512 + lane * 8 + i] = (__bf16)(v - (float)hi);
  }
}

// Every weight pack of a training step in ONE launch (the step re-packs ~47 matrices of a few KB each, ~6 us of launch
// latency apiece on a single stream): blockIdx.y = entry of a device-resident table, blockIdx.x * 256 + threadIdx.x = the
// thread index of the single-matrix kernels above / in conv_pw.hip (same layouts, bit-identical packs).
struct PrepackDesc { const float* w; void* wp; int Co, Ci, kind, trans; };   // kind: 0 fp32, 1 bf16, 2 bf16 hi + lo
static_assert(sizeof(PrepackDesc) == 32, "host side builds the table as 32-byte records");
__global__ __launch_bounds__(256) void pw_prepack_multi_kernel(const PrepackDesc* __restrict__ table) {
  const PrepackDesc e = table[blockIdx.y];
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int MT = (e.Co + 15) / 16;
  const float* __restrict__ w = e.w;
  if (e.kind == 0) {
    if (t >= (e.Ci / 4) * MT * 64) return;
    const int lane = t & 63, mt = (t >> 6) % MT, ks = (t >> 6) / MT;
    const int m = mt * 16 + (lane & 15), k = ks * 4 + (lane >> 4);
    float v = 0.0f;
    if (m < e.Co) v = e.trans ? w[(size_t)k * e.Co + m] : w[(size_t)m * e.Ci + k];
    reinterpret_cast<float*>(e.wp)[t] = v;
    return;
  }
  const int KK = (e.Ci + 31) / 32, NP2 = e.kind;
  if (t >= KK * MT * 64) return;
  __bf16* __restrict__ wp = reinterpret_cast<__bf16*>(e.wp);
  const int lane = t & 63, mt = (t >> 6) % MT, kk = (t >> 6) / MT;
  const int m = mt * 16 + (lane & 15), kb = kk * 32 + 8 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = kb + i;
    const float v = (m < e.Co && k < e.Ci) ? (e.trans ? w[(size_t)k * e.Co + m] : w[(size_t)m * e.Ci + k]) : 0.0f;
    const __bf16 hi = (__bf16)v;
    wp[((size_t)(kk * MT + mt) * NP2 + 0) * 512 + lane * 8 + i] = hi;
    if (NP2 == 2) wp[((size_t)(kk * MT + mt) * NP2 + 1) * 512 + lane * 8 + i] = (__bf16)(v - (float)hi);
  }
}

// TF: the conv input is act_in(tf_a[k] * x + tf_b[k]) evaluated on the way from LDS to the MFMA operand (training: the
// BatchNorm + activation of the depthwise conv fused into the project conv - the activated tensor is never written;
// models/mn/block_types.py:150-171 under model.train()); the SE scale (in_scale) multiplies the transformed value.
struct PwTf { const float* a; const float* b; int act; };

// XT / YT: storage type of x / y (act_io.h).  XT = bf16 (the wide tensors of the bf16-storage training plan): a chunk is
// staged as 32 rows x 512 B (one LDS-DMA instruction = two rows), a lane reads its 4 columns of 8 rows as 8-byte pieces and
// transposes them into B fragments with v_perm_b32 - no conversion, no lo part (host: NPROD = 1); x2 stays fp32 (the
// two-source GEMM of the expand data gradient reads the wide gradient g in bf16 and the narrow block input in fp32; host:
// c1 % 32 == 0, so a chunk is one or the other).
template <int MTW, int NPROD, int NSTG, bool TF, typename XT = float, typename YT = float>
__global__ __launch_bounds__(256, 2) void pw_conv_bf16_kernel(
    const XT* __restrict__ x, const __bf16* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ res, YT* __restrict__ y,
    float* __restrict__ pool, int B, int Ci, int Co, int S, int MT, int MC, int n_tiles, int NS, int act,
    int sc_bytes, int ci_x, PwTf tf, const float* __restrict__ x2, int c1, int tps, long long wp_bstride,
    float* __restrict__ stats, eat::PwGStat gs) {
  // gs.z != NULL (with stats): BatchNorm-backward partials of pw_epilogue_gstats (z in y's storage type) instead
  // tps > 0: per-sample weights (DyMN dynamic conv, models/dymn/dy_block.py:111-127, on split bf16 operands): a tile lies
  //          inside one sample (tps tiles per sample) and reads that sample's packed weights (wp_bstride elements apart)
  // x2 != NULL ("two-source"): the reduction axis is the channels of x (c1 rows) followed by the channels of x2
  // (Ci - c1 rows), both (B, *, S) - the data-gradient GEMM of the expand conv with its BatchNorm correction,
  // dx = [WaT | M] [g ; x] (train_fuse.hip), without a separate M x launch
  // ci_x: channels of x.  ci_x == Ci: plain 1x1 conv.  ci_x < Ci ("K-concat", DyMN): the reduction axis is nbank
  // copies of x's channels, k = bank * ci_x + ci - the weights are the banks side by side, the per-(sample, k) input
  // scale carries the attention (host: ci_x % 32 == 0, so a 32-row chunk never straddles two banks)
  constexpr int n_stages = NSTG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NP2 = NPROD == 3 ? 2 : 1;
  constexpr int kABytes = MTW * NP2 * 1024;                 // A fragments of one chunk
  constexpr bool XB = eat::Io<XT>::kBf;
  const int kXBytes = kKC * kTileN * ((XB && !x2) ? 2 : 4);
  const int kStage = kABytes + kXBytes + sc_bytes;           // + the SE scales of the chunk: kKC x NS floats, 256 B pieces
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
  const int mchunk = jj % MC, tile = (jj / MC) * 8 + xcd;
  if (tile >= n_tiles) return;
  const int mt0 = mchunk * MTW;
  // 32-bit index math: the host checks B*S < 2^31 (64-bit divisions cost ~150 VALU instructions each)
  const unsigned n_base = tps ? (unsigned)(tile / tps) * (unsigned)S + (unsigned)(tile % tps) * kTileN : (unsigned)tile * kTileN;
  const int b_first = (int)(n_base / (unsigned)S);
  const unsigned N = tps ? (unsigned)(b_first + 1) * (unsigned)S : (unsigned)B * (unsigned)S;   // first column this tile must not touch
  wp += tps ? (size_t)b_first * (size_t)wp_bstride : 0;
  unsigned nl = n_base + 4 * lane;
  if (nl > N - 4) nl = N - 4;
  const int bl = (int)(nl / (unsigned)S), sl = (int)(nl - (unsigned)bl * (unsigned)S);
  // XB: a lane moves 8 columns (16 bytes) of row (lane >> 5) of a row pair (host: S % 8 == 0)
  unsigned nl8 = n_base + 8 * (lane & 31);
  if (nl8 > N - 8) nl8 = N - 8;
  const int bl8 = (int)(nl8 / (unsigned)S), sl8 = (int)(nl8 - (unsigned)bl8 * (unsigned)S);
  const XT* xsrc = XB ? x + ((size_t)bl8 * (x2 ? c1 : ci_x)) * S + sl8 : x + ((size_t)bl * (x2 ? c1 : ci_x)) * S + sl;
  const float* xsrc2 = x2 ? x2 + ((size_t)bl * (Ci - c1)) * S + sl : nullptr;
  const unsigned nc = n_base + 64 * wv + 4 * (lane & 15);
  const bool col_ok = nc < N;
  const unsigned ncc = col_ok ? nc : N - 4;
  const int bc = (int)(ncc / (unsigned)S), sc_ = (int)(ncc - (unsigned)bc * (unsigned)S);
  const int kq = lane >> 4;
  const int n_chunks = (Ci + kKC - 1) / kKC;

  auto issue = [&](int c) {
    unsigned char* st = smem_raw + (c & (n_stages - 1)) * kStage;
    const int k0 = c * kKC;
    const int klen = (Ci - k0) < kKC ? (Ci - k0) : kKC;
    const int kx0 = k0 % ci_x;                                // row of x the chunk starts at
    float* Xs = reinterpret_cast<float*>(st + kABytes);
    bool rows16 = false;
    if constexpr (XB) rows16 = !(x2 && k0 >= c1);             // (block-uniform) bf16 rows: two per instruction
    if (rows16) {
#pragma unroll
      for (int i = 0; i < kKC / 8; ++i) {
        const int r = 2 * (wv + 4 * i) + (lane >> 5);
        const int rc = r < klen ? r : klen - 1;
        glds16_raw(xsrc + (size_t)(kx0 + rc) * S, reinterpret_cast<unsigned char*>(Xs) + (wv + 4 * i) * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < kKC / 4; ++i) {
        const int r = wv + 4 * i;
        const int rc = r < klen ? r : klen - 1;                // padded k: finite data x zero weight
        const int row = kx0 + rc;
        if constexpr (XB)
          glds16_raw(xsrc2 + (size_t)(row - c1) * S, Xs + r * kTileN);
        else
          glds16_raw((x2 && row >= c1) ? xsrc2 + (size_t)(row - c1) * S : xsrc + (size_t)row * S, Xs + r * kTileN);
      }
    }
#pragma unroll
    for (int i = 0; i < (MTW * NP2 + 3) / 4; ++i) {
      int q = wv + 4 * i;                                     // piece = (m-tile, hi/lo): 1 KiB
      if (q >= MTW * NP2) q = 0;
      int mt = mt0 + q / NP2;
      if (mt >= MT) mt = MT - 1;
      glds16_raw(wp + ((size_t)(c * MT + mt) * NP2 + (q % NP2)) * 512 + lane * 8, st + q * 1024);
    }
    if (in_scale) {
      float* SCs = reinterpret_cast<float*>(st + kABytes + kXBytes);
      for (int h = 0; h < (sc_bytes >> 8); ++h) {           // 2 pieces for planes >= 128 positions (NS <= 4)
        int e = h * 64 + lane;
        if (e >= kKC * NS) e = kKC * NS - 1;
        const int r = e / NS, j = e - r * NS;
        int bb = b_first + j;
        if (bb >= B) bb = B - 1;
        const int rc = r < klen ? r : klen - 1;
        glds4_raw(in_scale + (size_t)bb * Ci + k0 + rc, SCs + h * 64);
      }
    }
  };

  f32x4 acc[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  __shared__ float s_bias[128];
  eat::pw_stage_bias(bias, s_bias, mt0, Co, wv, lane);
  issue(0);
  for (int c = 0; c < n_chunks; ++c) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // chunk c landed; stage (c+1)&1 is free
    if (n_stages == 2 && c + 1 < n_chunks) issue(c + 1);
    const unsigned char* st = smem_raw + (c & (n_stages - 1)) * kStage;
    const float* Xw = reinterpret_cast<const float*>(st + kABytes) + (8 * kq) * kTileN + 64 * wv + 4 * (lane & 15);
    const float* SCs = reinterpret_cast<const float*>(st + kABytes + kXBytes) + (8 * kq) * NS + (bc - b_first);
    bf16x8 bh[4], bl_[4];
    bool raw16 = false;                                      // the chunk's rows are bf16 and need no arithmetic: transpose only
    float4 xr[8];
    bool rows16 = false;
    if constexpr (XB) rows16 = !(x2 && c * kKC >= c1);       // (block-uniform)
    if (rows16) {
      const unsigned char* Xb = st + kABytes + (8 * kq) * (kTileN * 2) + (64 * wv + 4 * (lane & 15)) * 2;
      u32x2 xq[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) xq[i] = *reinterpret_cast<const u32x2*>(Xb + i * (kTileN * 2));
      if (!TF && !in_scale) {
        raw16 = true;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          // column j of rows i, i + 1: the low (j even) / high (j odd) halves of dword j >> 1 of the two rows
          const unsigned p0 = __builtin_amdgcn_perm(xq[i + 1][0], xq[i][0], 0x05040100u), p1 = __builtin_amdgcn_perm(xq[i + 1][0], xq[i][0], 0x07060302u);
          const unsigned p2 = __builtin_amdgcn_perm(xq[i + 1][1], xq[i][1], 0x05040100u), p3 = __builtin_amdgcn_perm(xq[i + 1][1], xq[i][1], 0x07060302u);
          const bf16x2 h0 = __builtin_bit_cast(bf16x2, p0), h1 = __builtin_bit_cast(bf16x2, p1);
          const bf16x2 h2 = __builtin_bit_cast(bf16x2, p2), h3 = __builtin_bit_cast(bf16x2, p3);
          bh[0][i] = h0[0]; bh[0][i + 1] = h0[1]; bh[1][i] = h1[0]; bh[1][i + 1] = h1[1];
          bh[2][i] = h2[0]; bh[2][i + 1] = h2[1]; bh[3][i] = h3[0]; bh[3][i + 1] = h3[1];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          xr[i] = make_float4(eat::bf_lo(xq[i][0]), eat::bf_hi(xq[i][0]), eat::bf_lo(xq[i][1]), eat::bf_hi(xq[i][1]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[i] = *reinterpret_cast<const float4*>(Xw + i * kTileN);
    }
    if (!raw16) {
    if constexpr (TF) {
      // rows kb .. kb+7 of this lane (host: Ci % 8 == 0, so an octet is inside or outside as a whole; rows beyond Ci
      // meet zero weights: coefficient 0 keeps them finite)
      const int kb = c * kKC + 8 * kq;
      const bool in = kb < Ci;
      const float4 a0 = *reinterpret_cast<const float4*>(tf.a + (in ? kb : 0)), a1 = *reinterpret_cast<const float4*>(tf.a + (in ? kb : 0) + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(tf.b + (in ? kb : 0)), b1 = *reinterpret_cast<const float4*>(tf.b + (in ? kb : 0) + 4);
      const float ta[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float tb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const eat::ActCoef ac = eat::act_coef(tf.act);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float av = in ? ta[i] : 0.0f, bv = in ? tb[i] : 0.0f;
        xr[i].x = eat::act_apply(fmaf(av, xr[i].x, bv), ac); xr[i].y = eat::act_apply(fmaf(av, xr[i].y, bv), ac);
        xr[i].z = eat::act_apply(fmaf(av, xr[i].z, bv), ac); xr[i].w = eat::act_apply(fmaf(av, xr[i].w, bv), ac);
      }
    }
    if (in_scale) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float s = SCs[i * NS];
        xr[i].x *= s; xr[i].y *= s; xr[i].z *= s; xr[i].w *= s;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const float v0 = j == 0 ? xr[i].x : j == 1 ? xr[i].y : j == 2 ? xr[i].z : xr[i].w;
        const float v1 = j == 0 ? xr[i + 1].x : j == 1 ? xr[i + 1].y : j == 2 ? xr[i + 1].z : xr[i + 1].w;
        const bf16x2 h = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
        bh[j][i] = h[0]; bh[j][i + 1] = h[1];
        if constexpr (NPROD == 3) {
          const bf16x2 l = __builtin_convertvector(f32x2{v0 - (float)h[0], v1 - (float)h[1]}, bf16x2);
          bl_[j][i] = l[0]; bl_[j][i + 1] = l[1];
        }
      }
    }
    }
    const bf16x8* Af = reinterpret_cast<const bf16x8*>(st) + lane;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const bf16x8 ah = Af[(i * NP2) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
      if constexpr (NPROD == 3) {
        const bf16x8 al = Af[(i * NP2 + 1) * 64];
        // the three products of one accumulator are issued 4 MFMAs apart (back-to-back MFMAs on the SAME
        // accumulator stall on the read-after-write of the previous result)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl_[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
      }
    }
    if (n_stages == 1 && c + 1 < n_chunks) {      // single stage: refill it once every wave has consumed chunk c
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      issue(c + 1);
    }
  }

  if (stats && gs.z) {                                     // block-uniform; before the stores of y (see conv_pw.hip)
    __syncthreads();                                       // every wave is done with the operand stages: LDS is free
    eat::pw_epilogue_gstats<MTW, YT, YT>(acc, s_bias, reinterpret_cast<float*>(smem_raw), stats, gs, tile, mt0, kq, lane, wv,
                                         col_ok, bc, sc_, Co, S);
  }
  eat::pw_epilogue<MTW, YT>(acc, s_bias, res, y, pool, mt0, kq, lane, col_ok, bc, sc_, Co, S, act);
  if (stats && !gs.z) {                                    // train-mode statistics of the output (pw_epilogue.h)
    __syncthreads();
    eat::pw_epilogue_stats<MTW, YT>(acc, s_bias, reinterpret_cast<float*>(smem_raw), stats, tile, mt0, kq, lane, wv, col_ok, Co);
  }
}


template <int MTW, int NPROD, typename XT = float, typename YT = float>
int launch(hipStream_t s, const XT* x, const __bf16* wp, const float* bias, const float* in_scale, const float* res,
           YT* y, float* pool, int B, int Ci, int Co, int S, int MT, int MC, int act, int ci_x, PwTf tf,
           const float* x2, int c1, bool per_sample, float* stats, eat::PwGStat gs) {
  constexpr bool XB = eat::Io<XT>::kBf;
  const long long N = (long long)B * S;
  if (N > 0x7fff0000LL) return eat::fail(EAT_EINVAL, "eat_pw_conv_bf16_fwd: B*S = %lld exceeds the 32-bit column index", N);
  const int tps = per_sample ? (S + kTileN - 1) / kTileN : 0;
  const int n_tiles = per_sample ? B * tps : (int)((N + kTileN - 1) / kTileN);
  int NS = kTileN / S + 2;
  if (NS > B) NS = B;
  if (!in_scale) NS = 0;
  const int sc_bytes = in_scale ? ((kKC * NS + 63) / 64) * 256 : 0;
  constexpr int NP2 = NPROD == 3 ? 2 : 1;
  const long long wp_bstride = (long long)((Ci + kKC - 1) / kKC) * MT * NP2 * 512;
  // Two LDS stages (chunk c+1 in flight under the MFMAs of chunk c) when two such blocks fit a CU or when
  // there are not enough blocks for two per CU anyway; otherwise ONE stage, so that a second resident
  // block hides the load latency and the store phase instead (measured on MI355X, B=256: 80->480 95 vs
  // 135 us, 160->960 48 vs 83 us; the K-heavy 960->160 with 256 blocks keeps two stages: 45 vs 56 us).
  const size_t stage = (size_t)(MTW * NP2 * 1024 + kKC * kTileN * ((XB && !x2) ? 2 : 4) + sc_bytes);
  const int n_blocks = ((n_tiles + 7) / 8 * 8) * MC;
  const int n_stages = (2 * stage <= 78 * 1024 || n_blocks < 2 * 256) ? 2 : 1;
  const size_t smem = n_stages * stage;
  // (the on-load transform exists for fp32 -> fp32 and bf16 -> fp32 / bf16: the project conv of the training plans)
  constexpr bool TFOK = std::is_same<YT, float>::value || XB;
  if (tf.a && !TFOK) return eat::fail(EAT_EINVAL, "eat_pw_conv: the on-load transform needs an fp32 output or a bf16 input");
  auto kern = (tf.a && TFOK) ? (n_stages == 2 ? pw_conv_bf16_kernel<MTW, NPROD, 2, TFOK, XT, YT> : pw_conv_bf16_kernel<MTW, NPROD, 1, TFOK, XT, YT>)
                   : (n_stages == 2 ? pw_conv_bf16_kernel<MTW, NPROD, 2, false, XT, YT> : pw_conv_bf16_kernel<MTW, NPROD, 1, false, XT, YT>);
  if (smem > 160 * 1024) return eat::fail(EAT_EINVAL, "eat_pw_conv_bf16_fwd: LDS stage too large (%zu B; planes of %d positions)", smem, S);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eat::fail(EAT_ELAUNCH, "eat_pw_conv_bf16_fwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  const int tiles8 = (n_tiles + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(tiles8 * MC), dim3(256), smem, s, x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, MT, MC,
                     n_tiles, NS, act, sc_bytes, ci_x, tf, x2, c1, tps, wp_bstride, stats, gs);
  return eat::check_launch("eat_pw_conv_bf16_fwd");
}

template <int NPROD, typename XT = float, typename YT = float>
int dispatch(hipStream_t s, const XT* x, const __bf16* wp, const float* bias, const float* in_scale, const float* res,
             YT* y, float* pool, int B, int Ci, int Co, int S, int act, int ci_x, PwTf tf = PwTf{nullptr, nullptr, 0},
             const float* x2 = nullptr, int c1 = 0, bool per_sample = false, float* stats = nullptr,
             eat::PwGStat gs = eat::PwGStat{nullptr, nullptr, nullptr, 0}) {
  const int MT = (Co + 15) / 16;
  // (K-concat launches with few output rows - 128 x 1920 -> 320 @ 4x32: 192 blocks of 240 chunks - do NOT gain from more,
  // smaller row chunks: every block re-streams its x tile once per bank through L2, 425 -> 480 us with 448 blocks)
  const int MC = (MT + 7) / 8;
  const int mtw = (MT + MC - 1) / MC;
#define EAT_CASE(n) case n: return launch<n, NPROD, XT, YT>(s, x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, MT, (MT + n - 1) / n, act, ci_x, tf, x2, c1, per_sample, stats, gs);
  switch (mtw) {
    EAT_CASE(1) EAT_CASE(2) EAT_CASE(3) EAT_CASE(4) EAT_CASE(5) EAT_CASE(6) EAT_CASE(7) EAT_CASE(8)
    default: return eat::fail(EAT_EINVAL, "eat_pw_conv_bf16_fwd: internal tiling error");
  }
#undef EAT_CASE
}

}  // namespace

static int pw_prepack_bf16_impl(const float* w, const float* row_scale, void* wp, int Co, int Ci, int split, int trans,
                                eat_stream_t stream) {
  eat::clear_stale_error();
  const int MT = (Co + 15) / 16, KK = (Ci + 31) / 32;
  const int total = KK * MT * 64;
  hipLaunchKernelGGL(pw_prepack_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, row_scale,
                     reinterpret_cast<__bf16*>(wp), Co, Ci, MT, split ? 2 : 1, trans);
  return eat::check_launch("eat_pw_prepack_bf16");
}

extern "C" int eat_pw_prepack_bf16(const float* w, const float* row_scale, void* wp, int Co, int Ci, int split,
                                   eat_stream_t stream) {
  return pw_prepack_bf16_impl(w, row_scale, wp, Co, Ci, split, 0, stream);
}

extern "C" int eat_pw_prepack_bf16_t(const float* w_t, const float* row_scale, void* wp, int Co, int Ci, int split,
                                     eat_stream_t stream) {
  return pw_prepack_bf16_impl(w_t, row_scale, wp, Co, Ci, split, 1, stream);
}

extern "C" int eat_pw_prepack_multi(const void* table, int n, int max_threads, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!table || n < 1 || max_threads < 1) return eat::fail(EAT_EINVAL, "eat_pw_prepack_multi: empty table");
  hipLaunchKernelGGL(pw_prepack_multi_kernel, dim3((unsigned)((max_threads + 255) / 256), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const PrepackDesc*>(table));
  return eat::check_launch("eat_pw_prepack_multi");
}

extern "C" int eat_pw_conv_bf16_fwd(const float* x, const void* wp, const float* bias, const float* in_scale,
                                    const float* res, float* y, float* pool, int B, int Ci, int Co, int S, int act,
                                    int split, eat_stream_t stream) {
  eat::clear_stale_error();
  if (Ci % 4 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_bf16_fwd: Ci=%d must be a multiple of 4", Ci);
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_bf16_fwd: bad act %d", act);
  if (B < 1 || Co < 1 || S < 1) return eat::fail(EAT_EINVAL, "eat_pw_conv_bf16_fwd: bad shape");
  const __bf16* w16 = reinterpret_cast<const __bf16*>(wp);
  hipStream_t s = (hipStream_t)stream;
  if (S % 4 != 0)      // planes that do not start on 16-byte boundaries: plain 4-byte kernel on the same packs
    return eat::pw_conv_generic(x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, act, split ? 2 : 1, 0, s);
  {
    const int rc = eat::pw_stream_try(x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, act, split, Ci, s);
    if (rc != 1) return rc;
  }
  return split ? dispatch<3>(s, x, w16, bias, in_scale, res, y, pool, B, Ci, Co, S, act, Ci)
               : dispatch<1>(s, x, w16, bias, in_scale, res, y, pool, B, Ci, Co, S, act, Ci);
}

// 1x1 conv whose input is act_in(tf_a[k] x + tf_b[k]) [* in_scale[b,k]] evaluated on load (see PwTf above)
namespace eat {
int pw_conv_bf16_tf(const float* x, const float* tf_a, const float* tf_b, int tf_act, const void* wp, const float* bias,
                    const float* in_scale, const float* res, float* y, int B, int Ci, int Co, int S, int act, int split,
                    hipStream_t s) {
  const PwTf tf{tf_a, tf_b, tf_act};
  const __bf16* w16 = reinterpret_cast<const __bf16*>(wp);
  return split ? dispatch<3>(s, x, w16, bias, in_scale, res, y, nullptr, B, Ci, Co, S, act, Ci, tf)
               : dispatch<1>(s, x, w16, bias, in_scale, res, y, nullptr, B, Ci, Co, S, act, Ci, tf);
}
}  // namespace eat

// train-mode conv z = W x with the statistics epilogue (eat_pw_conv_stats_fwd, conv_pw.hip)
namespace eat {
int pw_conv_bf16_stats(const float* x, const void* wp, int split, int per_sample, const float* tf_a, const float* tf_b,
                       int tf_act, const float* in_scale, const float* zero_bias, float* y, float* part, int B, int Ci, int Co,
                       int S, hipStream_t s, PwGStat gs) {
  const PwTf tf{tf_a, tf_b, tf_act};
  const __bf16* w16 = reinterpret_cast<const __bf16*>(wp);
  return split ? dispatch<3>(s, x, w16, zero_bias, in_scale, nullptr, y, nullptr, B, Ci, Co, S, EAT_ACT_NONE, Ci, tf, nullptr, 0,
                             per_sample != 0, part, gs)
               : dispatch<1>(s, x, w16, zero_bias, in_scale, nullptr, y, nullptr, B, Ci, Co, S, EAT_ACT_NONE, Ci, tf, nullptr, 0,
                             per_sample != 0, part, gs);
}
}  // namespace eat

// 1x1 conv over the concatenated channels of two tensors (see the kernel's x2 / c1)
namespace eat {
int pw_conv_bf16_cat(const float* x1, int c1, const float* x2, int c2, const void* wp, const float* bias, const float* res,
                     float* y, int B, int Co, int S, int act, int split, hipStream_t s) {
  const __bf16* w16 = reinterpret_cast<const __bf16*>(wp);
  const PwTf none{nullptr, nullptr, 0};
  return split ? dispatch<3>(s, x1, w16, bias, nullptr, res, y, nullptr, B, c1 + c2, Co, S, act, c1 + c2, none, x2, c1)
               : dispatch<1>(s, x1, w16, bias, nullptr, res, y, nullptr, B, c1 + c2, Co, S, act, c1 + c2, none, x2, c1);
}
}  // namespace eat

// DyMN dynamic 1x1 conv with per-sample weights on split bf16 operands: wp_b = eat_dyn_pw_pack_bf16's (B, KK*MT*2*512)
// hi / lo fragments (models/dymn/dy_block.py:111-127; the fp32-fragment form is eat_pw_conv_dyn_fwd).  Ci % 4 == 0, S % 4 == 0.
extern "C" int eat_pw_conv_dyn_bf16_fwd(const float* x, const void* wp_b, const float* bias, const float* res, float* y,
                                        int B, int Ci, int Co, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (Ci % 4 != 0 || S % 4 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_bf16_fwd: Ci=%d and S=%d must be multiples of 4", Ci, S);
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_bf16_fwd: bad act %d", act);
  if (B < 1 || Co < 1 || !wp_b) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_bf16_fwd: bad arguments");
  return dispatch<3>((hipStream_t)stream, x, reinterpret_cast<const __bf16*>(wp_b), bias, nullptr, res, y, nullptr, B, Ci, Co,
                     S, act, Ci, PwTf{nullptr, nullptr, 0}, nullptr, 0, true);
}

// DyMN dynamic 1x1 conv WITHOUT per-sample weights (models/dymn/dy_block.py:103-131):
//   z_b = (sum_k att[b,k] W_k) x_b = [W_0 | ... | W_{K-1}] [att[b,0] x_b ; ... ; att[b,K-1] x_b]
// one GEMM over the K-concatenated banks (wp: eat_pw_prepack_bf16 of the Co x (nbank * Ci) matrix, shared by every
// sample) with the attention as the per-(sample, k) input scale (att_scale: (B, nbank * Ci), att[b,k] repeated Ci times).
extern "C" int eat_pw_conv_kcat_fwd(const float* x, const void* wp, const float* bias, const float* att_scale,
                                    const float* res, float* y, int B, int Ci, int nbank, int Co, int S, int act,
                                    eat_stream_t stream) {
  eat::clear_stale_error();
  if (Ci % 32 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_kcat_fwd: Ci=%d must be a multiple of 32", Ci);
  if (S % 4 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_kcat_fwd: S=%d must be a multiple of 4", S);
  if (nbank < 1 || B < 1 || Co < 1 || !att_scale) return eat::fail(EAT_EINVAL, "eat_pw_conv_kcat_fwd: bad arguments");
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_kcat_fwd: bad act %d", act);
  {
    const int rc = eat::pw_stream_try(x, wp, bias, att_scale, res, y, nullptr, B, nbank * Ci, Co, S, act, 1, Ci,
                                      (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  return dispatch<3>((hipStream_t)stream, x, reinterpret_cast<const __bf16*>(wp), bias, att_scale, res, y, nullptr, B,
                     nbank * Ci, Co, S, act, Ci);
}

// ---- 1x1 conv of the bf16-STORAGE training plan (act_io.h; BASELINE configs[2], the reference's 16-bit mixed precision,
// ex_pl_audioset.py:287-293 over models/mn/block_types.py:138-147,167-181): plain bf16 operands, fp32 accumulation, and
// the WIDE tensor of the layer - the expand conv's output, the project conv's input, the project data gradient's output,
// the expand data gradient's input - in bf16 in HBM, the narrow one in fp32.
//   x_b16 = 0, y_b16 = 1   z_e = W x (expand conv) / dxs = Wp^T dz_p (project data gradient): no transform, scale, residual
//   x_b16 = 1, y_b16 = 0   project conv z_p = Wp act(tf_a x + tf_b) * in_scale with the statistics epilogue (stats_part, as
//                          eat_pw_conv_stats_fwd), or the two-source data-gradient GEMM dx = [WaT | M] [g ; x2] + bias + res
//                          (x2 fp32 with Ci - c1 channels, c1 % 32 == 0; as eat_pw_conv_cat_fwd)
//   x_b16 = 1, y_b16 = 1   the project conv with z_p stored in bf16 as well (statistics of the stored values); no x2 / res
// wp: eat_pw_prepack_bf16(split = 0) of the (Co, Ci) matrix.  S % 8 == 0, Ci % 4 == 0 (% 8 with a transform).
extern "C" int eat_pw_conv_b16_fwd(const void* x, int x_b16, const float* x2, int c1, const void* wp, const float* bias,
                                   const float* tf_a, const float* tf_b, int tf_act, const float* in_scale, const float* res,
                                   void* y, int y_b16, float* stats_part, const void* gz, const float* g_a, const float* g_b,
                                   int g_act, int B, int Ci, int Co, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !wp || !bias || !y) return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: missing operand");
  if (B < 1 || Co < 1 || Ci < 4 || Ci % 4 != 0 || S < 8 || S % 8 != 0)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: Ci=%d must be a multiple of 4 and S=%d a multiple of 8", Ci, S);
  if (act < 0 || act > 2 || tf_act < 0 || tf_act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: bad act");
  if ((tf_a == nullptr) != (tf_b == nullptr)) return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: tf_a and tf_b go together");
  if (tf_a && Ci % 8 != 0) return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: the on-load transform needs Ci %% 8 == 0 (Ci=%d)", Ci);
  const __bf16* w16 = reinterpret_cast<const __bf16*>(wp);
  const PwTf tf{tf_a, tf_b, tf_act};
  hipStream_t s = (hipStream_t)stream;
  if (!x_b16 && y_b16) {
    if (x2 || tf_a || in_scale || res || stats_part)
      return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: a bf16 output takes a plain conv only");
    return dispatch<1, float, eat::bf16_t>(s, reinterpret_cast<const float*>(x), w16, bias, nullptr, nullptr,
                                           reinterpret_cast<eat::bf16_t*>(y), nullptr, B, Ci, Co, S, act, Ci);
  }
  if (x_b16 && !y_b16) {
    if (x2 && (c1 < 32 || c1 % 32 != 0 || c1 >= Ci || tf_a || in_scale || stats_part))
      return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: two-source form needs c1 %% 32 == 0 (c1=%d) and no transform / scale / statistics", c1);
    return dispatch<1, eat::bf16_t, float>(s, reinterpret_cast<const eat::bf16_t*>(x), w16, bias, in_scale, res,
                                           reinterpret_cast<float*>(y), nullptr, B, Ci, Co, S, act, Ci, tf, x2, c1, false,
                                           stats_part);
  }
  if (gz && !(x_b16 && y_b16 && stats_part && g_a && g_b && g_act >= 0 && g_act <= 2 && !tf_a && !in_scale))
    return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: the BatchNorm-backward epilogue (gz) goes with a plain bf16 -> bf16 conv and stats_part");
  if (x_b16 && y_b16) {
    if (x2 || res) return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: bf16 -> bf16 takes no second source / residual");
    // gz != NULL: stats_part receives the BatchNorm-backward partials of pw_epilogue_gstats (gz = the bf16 z_d) instead
    return dispatch<1, eat::bf16_t, eat::bf16_t>(s, reinterpret_cast<const eat::bf16_t*>(x), w16, bias, in_scale, nullptr,
                                                 reinterpret_cast<eat::bf16_t*>(y), nullptr, B, Ci, Co, S, act, Ci, tf, nullptr, 0,
                                                 false, stats_part, eat::PwGStat{gz, g_a, g_b, g_act});
  }
  return eat::fail(EAT_EINVAL, "eat_pw_conv_b16_fwd: at least one of x / y is a bf16 tensor");
}

// ---- per-sample-weight 1x1 conv of the bf16-STORAGE plan for DyMN (models/dymn/dy_block.py:103-131 with a 1x1 kernel, under the
// reference's 16-bit mixed precision, ex_pl_audioset.py:287-293): plain bf16 operands, fp32 accumulation, the WIDE tensor of the
// layer in bf16 in HBM.  wp_b: eat_dyn_pw_pack_b16 (one plain-bf16 pack per sample; tiles lie inside samples).
//   x_b16 = 0, y_b16 = 1   dynamic expand conv z_e = W_b x (stats_part != NULL: batch statistics of the STORED z_e in the epilogue,
//                          tiles = eat_pw_conv_stat_tiles(B, S, 1)), or the project data gradient dx2 = W_b^T dz_p
//   x_b16 = 1, y_b16 = 0   dynamic project conv z_p = W_b x2 (+ statistics), or the expand data gradient dx = W_b^T dz_e + res
// S % 8 == 0, Ci % 4 == 0.
extern "C" int eat_pw_conv_dyn_b16_fwd(const void* x, int x_b16, const void* wp_b, const float* bias, const float* res, void* y,
                                       int y_b16, float* stats_part, int B, int Ci, int Co, int S, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!x || !wp_b || !bias || !y) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_b16_fwd: missing operand");
  if (B < 1 || Co < 1 || Ci < 4 || Ci % 4 != 0 || S < 8 || S % 8 != 0)
    return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_b16_fwd: Ci=%d must be a multiple of 4 and S=%d a multiple of 8", Ci, S);
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_b16_fwd: bad act %d", act);
  if ((x_b16 != 0) == (y_b16 != 0)) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_b16_fwd: exactly one of x / y is the bf16 (wide) tensor");
  const __bf16* w16 = reinterpret_cast<const __bf16*>(wp_b);
  const PwTf none{nullptr, nullptr, 0};
  hipStream_t s = (hipStream_t)stream;
  if (y_b16) {
    if (res) return eat::fail(EAT_EINVAL, "eat_pw_conv_dyn_b16_fwd: a bf16 output takes no residual");
    return dispatch<1, float, eat::bf16_t>(s, reinterpret_cast<const float*>(x), w16, bias, nullptr, nullptr,
                                           reinterpret_cast<eat::bf16_t*>(y), nullptr, B, Ci, Co, S, act, Ci, none, nullptr, 0, true,
                                           stats_part);
  }
  return dispatch<1, eat::bf16_t, float>(s, reinterpret_cast<const eat::bf16_t*>(x), w16, bias, nullptr, res,
                                         reinterpret_cast<float*>(y), nullptr, B, Ci, Co, S, act, Ci, none, nullptr, 0, true,
                                         stats_part);
}
