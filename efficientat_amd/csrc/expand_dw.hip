// Expand 1x1 conv + BN + act -> depthwise 3x3 conv + BN + act (+ SE squeeze sums) in ONE kernel for the small late-layer
// planes (models/mn/block_types.py:138-162 and the squeeze of :72-73; mn10 blocks 8-12: 8 x 63 planes, C_in 80 / 112,
// C_exp 184 ... 672).  The expanded tensor - the largest tensor of the block - is never written to or read from HBM:
// for 112 -> 672 that is 2 x 1.35 MB of the 3.1 MB the two separate kernels move per clip.
//
// One block (8 waves) = one sample (or 1/SPLIT of its channel groups when the batch alone does not fill the chip); it loads
// and splits the sample's x ONCE and then loops over groups of 64 expanded channels:
//   phase 1  = pw_expand_kernel's data movement (conv_pw_stream.hip): wave w owns the columns [64 w, 64 w + 64) of the
//              sample's plane (S = F*T <= 512 positions); it loads them for ALL C_in rows straight into the MFMA
//              B-operand layout (16-byte loads), splits them ONCE into bf16 hi / lo (bf16x3 arithmetic, the packs of
//              eat_pw_prepack_bf16) and walks down the block's 4 m-tiles: accumulators start at the bias, A fragments
//              by raw buffer loads one m-tile ahead, activation, and the 16 x 64 result goes to an LDS tile
//              E[channel][position] (16-byte ds_write, 130 KB for 64 channels of 504 positions) instead of HBM.
//   barrier
//   phase 2  = dw_plane_kernel's register scheme (dw_plane.hip) fed from LDS: wave w takes the planes w, w + 8, ...;
//              lane t holds column t of all F rows (conflict-free 4-byte LDS reads), horizontal neighbours by DPP
//              wavefront shifts (lane 0 / lane >= T read 0 = the zero padding), vertical taps are other registers, the 9
//              taps and the bias are wave-uniform scalars; activation, coalesced row stores, one atomicAdd of the plane
//              sum for the squeeze-excitation mean.
//   barrier (the tile is free again); the row stores of phase 2 drain under the MFMAs of the next group's phase 1.
// The blocks of one sample run on one XCD (its L2 serves the second read of x when SPLIT > 1).
#include <cstdlib>
#include "eat_common.h"
#include "pw_epilogue.h"
#include "bf16_frag.h"

namespace {

using namespace eatfrag;

constexpr int kGroupTiles = 4;                            // m-tiles (of 16 channels) per block
constexpr int kMaxRows = 8;                               // plane rows held in registers in phase 2

template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_prev(float v) { return dpp0<0x138>(v); }   // wave_shr:1, lane 0 reads 0
__device__ __forceinline__ float from_next(float v) { return dpp0<0x130>(v); }   // wave_shl:1, lane 63 reads 0

// wave sum by DPP adds (dymn.hip): scan inside each 16-lane row, row_bcast:15 / :31 across rows; lane 63 holds the total
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

template <int NCH>
__global__ __launch_bounds__(512, 1) void expand_dw_kernel(
    const float* __restrict__ x, const bf16x8* __restrict__ wp, const float* __restrict__ bias_e,
    const float* __restrict__ w_d, const float* __restrict__ bias_d, float* __restrict__ y, float* __restrict__ pool,
    int B, int Ci, int Ce, int F, int T, int MT, int NG, int SPLIT, int SP, int act) {
  constexpr int NP2 = 2;                                    // bf16x3: hi and lo parts
  extern __shared__ __attribute__((aligned(16))) float smem[];   // E[kGroupTiles * 16][SP], then the bias of every row
  float* E = smem;
  float* s_bias = smem + (size_t)kGroupTiles * 16 * SP;     // [MT * 16]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the blocks of one sample run on one XCD
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
  const int b = (jj / SPLIT) * 8 + xcd, part = jj % SPLIT;
  if (b >= B) return;
  const int S = F * T;
  for (int e = tid; e < MT * 16; e += 512) s_bias[e] = bias_e[e < Ce ? e : Ce - 1];

  // ------------------------------------------------------------------ x -> B fragments, once per block
  const int kq = lane >> 4;
  const int sc = 64 * wv + 4 * (lane & 15);                 // the lane's 4 consecutive positions
  const bool col_ok = sc < S;                               // S % 4 == 0: a quad is inside or outside as a whole
  bf16x8 bh[NCH][4], bl[NCH][4];
  {
    const float* xcol = x + (size_t)b * Ci * S + (col_ok ? sc : S - 4);
    auto load_rows = [&](int c, float4 (&xr)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = c * kKC + 8 * kq + i;
        xr[i] = *reinterpret_cast<const float4*>(xcol + (size_t)(k < Ci ? k : Ci - 1) * S);   // padded k: finite x, zero w
      }
    };
    float4 xa[8], xb[8];
    load_rows(0, xa);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) {
        if ((c & 1) == 0) load_rows(c + 1, xb); else load_rows(c + 1, xa);
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((c & 1) == 0) split_rows<3>(xa, bh[c], bl[c]); else split_rows<3>(xb, bh[c], bl[c]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (long long)NCH * MT * NP2 * 1024);
  const unsigned a_voff = lane * 16;
  auto frag_soff = [&](int c, int mt, int h) { return (unsigned)(((c * MT + mt) * NP2 + h) * 1024); };
  const eat::ActCoef ac = eat::act_coef(act);
  __syncthreads();                                          // s_bias is complete

  // A fragments: two register sets alternate (conv_pw_stream.hip).  Every group but the last has 4 m-tiles, so a group
  // starts and ends on set a0 - the last m-tile of a group requests the FIRST m-tile of the block's next group, whose
  // fragments then arrive under phase 2.
  bf16x8 a0[NCH][NP2], a1[NCH][NP2];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int h = 0; h < NP2; ++h) a0[c][h] = buf_load_frag(rw, a_voff, frag_soff(c, part * kGroupTiles, h));

  for (int cg = part; cg < NG; cg += SPLIT) {
    const int mt0 = cg * kGroupTiles;
    const int mt1 = (mt0 + kGroupTiles) < MT ? (mt0 + kGroupTiles) : MT;
    const int next_first = (cg + SPLIT < NG) ? (cg + SPLIT) * kGroupTiles : -1;
    // ---------------------------------------------------------------- phase 1: expand into the LDS tile
    {
      auto m_tile = [&](int mt, const bf16x8 (&cur)[NCH][NP2], bf16x8 (&nxt)[NCH][NP2]) {
        const int mtn = (mt + 1 < mt1) ? mt + 1 : (next_first >= 0 ? next_first : mt);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int h = 0; h < NP2; ++h) nxt[c][h] = buf_load_frag(rw, a_voff, frag_soff(c, mtn, h));
        const f32x4 bv = *reinterpret_cast<const f32x4*>(s_bias + mt * 16 + 4 * kq);
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = bv;             // acc[j][r] = bias of row 4 * kq + r
#pragma unroll
        for (int c = 0; c < NCH; ++c) mfma_chunk<3>(acc, cur[c][0], cur[c][1], bh[c], bl[c]);
        float* erow = E + (size_t)((mt - mt0) * 16 + 4 * kq) * SP + sc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 v = {eat::act_apply(acc[0][r], ac), eat::act_apply(acc[1][r], ac), eat::act_apply(acc[2][r], ac),
                           eat::act_apply(acc[3][r], ac)};
          if (col_ok) *reinterpret_cast<f32x4*>(erow + (size_t)r * SP) = v;
        }
      };
      int mt = mt0;
      for (; mt + 1 < mt1; mt += 2) {
        m_tile(mt, a0, a1);
        m_tile(mt + 1, a1, a0);
      }
      if (mt < mt1) m_tile(mt, a0, a1);
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase 2: depthwise 3x3 on the planes of the tile
    {
      const int t = lane;
      const bool t_ok = t < T;
      // column validity as a 0 / 1 factor the optimiser cannot see through, row stores through a range-checked buffer
      // descriptor: with `(t_ok && f < F) ? E[..] : 0` and `if (t_ok) store` hipcc put each of the 8 LDS row reads of a plane
      // into its own exec-masked block with its own s_waitcnt, and a branch around every row store (round-2 ISA finding)
      const float t_mask = eat::opaque(t_ok ? 1.0f : 0.0f);
      const unsigned t_off = t_ok ? 4u * (unsigned)t : 0x80000000u;
      const int n_planes = (mt1 - mt0) * 16;
      // two planes per trip: the scalar tap loads and the LDS reads of one overlap the FMAs of the other (more would not
      // fit next to the 128 registers of x fragments when C_in > 96).  (A packed-fp32 version - the two planes as the
      // halves of v_pk_fma_f32 operands - needs ~100 v_mov to build the pairs, has no packed max / min for the
      // activation, and spills at C_in > 96: not kept.)
#pragma unroll 1
      for (int i0 = 0; i0 < 8; i0 += 2) {
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
          const int p = wv + 8 * i;                           // wave-uniform
          const int c = mt0 * 16 + p;
          if (p < n_planes && c < Ce) {                       // rows of a short / ragged last group
            float wt[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) wt[q] = w_d[(size_t)c * 9 + q];
            const float bd = bias_d[c];
            const float* ep = E + (size_t)p * SP;
            float e[kMaxRows], l[kMaxRows], r[kMaxRows];
#pragma unroll
            for (int f = 0; f < kMaxRows; ++f) {
              int idx = f * T + t;
              if (idx > S - 1) idx = S - 1;                   // masked below; keeps the address inside the plane
              const float v = ep[idx];
              e[f] = v * (t_mask * eat::opaque(f < F ? 1.0f : 0.0f));
            }
#pragma unroll
            for (int f = 0; f < kMaxRows; ++f) { l[f] = from_prev(e[f]); r[f] = from_next(e[f]); }
            float psum = 0.0f;
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + ((size_t)b * Ce + c) * S, 0, 4 * S, 0x00020000);
#pragma unroll
            for (int f = 0; f < kMaxRows; ++f) {
              if (f < F) {                                    // wave-uniform
                float o = bd;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                  const int ff = f + u - 1;
                  if (ff >= 0 && ff < kMaxRows) {             // rows outside the plane are zero padding (e = 0 for ff >= F)
                    o = fmaf(wt[u * 3 + 0], l[ff], o);
                    o = fmaf(wt[u * 3 + 1], e[ff], o);
                    o = fmaf(wt[u * 3 + 2], r[ff], o);
                  }
                }
                o = eat::act_apply(o, ac);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), yr, (int)t_off, 4 * f * T, 0);
                psum = fmaf(o, t_mask, psum);
              }
            }
            if (pool) {                                       // DPP adds at VALU rate (no LDS crossbar): total in lane 63
              psum = wave_total_in_lane63(psum);
              if (lane == 63) atomicAdd(pool + (size_t)b * Ce + c, psum);
            }
          }
        }
      }
    }
    __syncthreads();                                        // every wave is done with the tile before it is overwritten
  }
}

template <int NCH>
int launch(hipStream_t s, const float* x, const void* wp, const float* bias_e, const float* w_d, const float* bias_d,
           float* y, float* pool, int B, int Ci, int Ce, int F, int T, int act) {
  const int MT = (Ce + 15) / 16;
  const int NG = (MT + kGroupTiles - 1) / kGroupTiles;
  const int S = F * T;
  const int SP = S + 4;
  // one block per sample when the batch fills the chip (x is loaded and split once per sample); smaller batches split
  // the channel groups of a sample over several blocks
  int SPLIT = (256 + B - 1) / B;
  if (SPLIT > NG) SPLIT = NG;
  if (SPLIT < 1) SPLIT = 1;
  const size_t smem = ((size_t)kGroupTiles * 16 * SP + (size_t)MT * 16) * sizeof(float);
  auto kern = expand_dw_kernel<NCH>;
  if (smem > 160 * 1024) return eat::fail(EAT_EINVAL, "eat_expand_dw_bf16_fwd: LDS tile too large (%zu B)", smem);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eat::fail(EAT_ELAUNCH, "eat_expand_dw_bf16_fwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  const int grid = ((B + 7) / 8) * 8 * SPLIT;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, x, reinterpret_cast<const bf16x8*>(wp), bias_e, w_d, bias_d, y,
                     pool, B, Ci, Ce, F, T, MT, NG, SPLIT, SP, act);
  return eat::check_launch("eat_expand_dw_bf16_fwd");
}

}  // namespace

// x (B, Ci, F, T) -> y (B, Ce, F, T) = act(dw3x3(act(W_e x + b_e)) + b_d); wp_e: eat_pw_prepack_bf16(split = 1) of the
// BN-folded expand weights, w_d (Ce, 9) / bias_d (Ce) the BN-folded depthwise taps; pool (B, Ce) or NULL accumulates
// the plane sums of y.  Geometry: k = 3, stride 1, T <= 64, F <= 8, F*T a multiple of 4 and <= 512, Ci <= 128 and a
// multiple of 4 - anything else is EAT_EINVAL (the caller keeps eat_pw_conv_bf16_fwd + eat_dw_conv_fwd for it).
extern "C" int eat_expand_dw_bf16_fwd(const float* x, const void* wp_e, const float* bias_e, const float* w_d,
                                      const float* bias_d, float* y, float* pool, int B, int Ci, int Ce, int F, int T,
                                      int k, int stride, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  const int S = F * T;
  if (k != 3 || stride != 1 || T < 1 || T > 64 || F < 1 || F > kMaxRows || S % 4 != 0 || S > 512 || Ci % 4 != 0 || Ci < 4 ||
      Ci > 128 || B < 1 || Ce < 1)
    return eat::fail(EAT_EINVAL, "eat_expand_dw_bf16_fwd: unsupported geometry (k=%d stride=%d F=%d T=%d Ci=%d)", k, stride, F, T, Ci);
  if (act < 0 || act > 2) return eat::fail(EAT_EINVAL, "eat_expand_dw_bf16_fwd: bad act %d", act);
  hipStream_t s = (hipStream_t)stream;
  switch ((Ci + 31) / 32) {
    case 1: return launch<1>(s, x, wp_e, bias_e, w_d, bias_d, y, pool, B, Ci, Ce, F, T, act);
    case 2: return launch<2>(s, x, wp_e, bias_e, w_d, bias_d, y, pool, B, Ci, Ce, F, T, act);
    case 3: return launch<3>(s, x, wp_e, bias_e, w_d, bias_d, y, pool, B, Ci, Ce, F, T, act);
    default: return launch<4>(s, x, wp_e, bias_e, w_d, bias_d, y, pool, B, Ci, Ce, F, T, act);
  }
}
