// Shared epilogue of the 1x1-convolution kernels (conv_pw.hip, conv_pw_bf16.hip):
//   y = act(acc + bias) [+ residual], optional plane sums (SE squeeze / global average pool).
// Replaces the tail of models/mn/block_types.py:138-147,167-171,177-181 (BN affine folded into the
// weights, activation, residual add) and the pooling of models/mn/model.py:220.
//
// Written so that the stores stream: gfx9 counts loads and stores on the same in-order vmcnt, so a
// load whose result is needed right away also waits for every store issued before it.  The bias
// values of the block's rows are therefore staged in LDS before the main loop (pw_stage_bias: an
// LDS-DMA that rides in front of the first k-chunk), the activation is branch-free (uniform
// coefficients instead of a switch), and the residual of m-tile i+1 is requested before the stores of
// m-tile i are issued.
#pragma once
#include "eat_common.h"
#include "act_io.h"

namespace eat {

using acc_f32x4 = __attribute__((ext_vector_type(4))) float;

struct ActCoef { float lo, a, b; };

// act(v) = max(v, lo) * clamp(v*a + b, 0, 1):  none (-inf, 0, 1), ReLU (0, 0, 1), Hardswish (-inf, 1/6, 1/2)
__device__ __forceinline__ ActCoef act_coef(int act) {
  ActCoef c;
  c.lo = act == EAT_ACT_RELU ? 0.0f : -__builtin_huge_valf();
  c.a = act == EAT_ACT_HSWISH ? (1.0f / 6.0f) : 0.0f;
  c.b = act == EAT_ACT_HSWISH ? 0.5f : 1.0f;
  return c;
}
__device__ __forceinline__ float act_apply(float v, const ActCoef& c) {
  return fmaxf(v, c.lo) * __builtin_amdgcn_fmed3f(fmaf(v, c.a, c.b), 0.0f, 1.0f);
}

// Every wave DMAs 64 of the block's (at most 128) bias values into s_bias; issue it BEFORE the first
// k-chunk so that the chunk's own wait + barrier also publishes the bias.
typedef __attribute__((address_space(3))) void epi_lds_void;
__device__ __forceinline__ void pw_stage_bias(const float* __restrict__ bias, float* s_bias, int mt0, int Co, int wv,
                                              int lane) {
  const int half = wv & 1;
  int m = mt0 * 16 + half * 64 + lane;
  if (m >= Co) m = Co - 1;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(epi_lds_void*)(s_bias + half * 64));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(bias + m) : "memory", "m0");
}

// Lane layout (both kernels): lane owns rows m = (mt0+i)*16 + kq*4 + r (i < MTW, r < 4) and the 4
// consecutive columns starting at position sc_ of sample bc; acc[i][j][r] is column j of row r.
// YT: storage type of y (act_io.h); the pooled variant and the residual are fp32-only (host checks)
template <int MTW, bool LINEAR, typename YT = float>
__device__ __forceinline__ void pw_epilogue_act(const acc_f32x4 (&acc)[MTW][4], const float* s_bias,
                                                const float* __restrict__ res, YT* __restrict__ y,
                                                float* __restrict__ pool, int mt0, int kq, int lane, bool col_ok,
                                                int bc, int sc_, int Co, int S, const ActCoef ac) {
  const size_t plane = (size_t)S;
  const size_t base = (size_t)bc * Co * plane + sc_;
  auto value = [&](int i, int r) {
    const float bm = s_bias[i * 16 + kq * 4 + r];
    if constexpr (LINEAR)
      return make_float4(acc[i][0][r] + bm, acc[i][1][r] + bm, acc[i][2][r] + bm, acc[i][3][r] + bm);
    else
      return make_float4(act_apply(acc[i][0][r] + bm, ac), act_apply(acc[i][1][r] + bm, ac),
                         act_apply(acc[i][2][r] + bm, ac), act_apply(acc[i][3][r] + bm, ac));
  };
  auto row_of = [&](int i, int r) { return (mt0 + i) * 16 + kq * 4 + r; };
  auto store_tile = [&](int i, const float4 (&v)[4]) {
    const bool full = (mt0 + i + 1) * 16 <= Co;                      // wave-uniform
    YT* yr = y + base + (size_t)row_of(i, 0) * plane;
    if (full) {
      if (col_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Io<YT>::store4(yr + (size_t)r * plane, v[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col_ok && row_of(i, r) < Co) Io<YT>::store4(yr + (size_t)r * plane, v[r]);
    }
  };

  if (!pool) {
    if (res) {
      float4 rv[4], rn[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = row_of(0, r);
        rv[r] = *reinterpret_cast<const float4*>(res + base + (size_t)(m < Co ? m : Co - 1) * plane);
      }
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        if (i + 1 < MTW) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = row_of(i + 1, r);
            rn[r] = *reinterpret_cast<const float4*>(res + base + (size_t)(m < Co ? m : Co - 1) * plane);
          }
        }
        // values first (no branch around the activation math), then ONE exec-masked region per m-tile: a tile whose 16
        // rows all exist (wave-uniform test) stores under the column mask only; the ragged last tile adds the row test.
        // (With `if (col_ok && m < Co) store(value)` per row hipcc emitted a saveexec / branch pair around every store
        // AND its activation math: 32 of them per wave at 8 m-tiles.)
        float4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = value(i, r);
          v[r].x += rv[r].x; v[r].y += rv[r].y; v[r].z += rv[r].z; v[r].w += rv[r].w;
        }
        store_tile(i, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = rn[r];
      }
    } else {
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        float4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = value(i, r);
        store_tile(i, v);
      }
    }
    return;
  }

  // pooled variant (head conv: y may be NULL, the plane sums are the product)
  const int b_lo = __shfl(bc, lane & ~15, 64), b_hi = __shfl(bc, lane | 15, 64);
  const bool group_one_sample = (b_lo == b_hi);
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = row_of(i, r);
      const bool ok = col_ok && m < Co;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        v = value(i, r);
        const size_t off = base + (size_t)m * plane;
        if (res) {
          const float4 q = *reinterpret_cast<const float4*>(res + off);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (y) Io<YT>::store4(y + off, v);
      }
      float ps = v.x + v.y + v.z + v.w;
      if (group_one_sample) {
        ps += __shfl_xor(ps, 1, 64); ps += __shfl_xor(ps, 2, 64);
        ps += __shfl_xor(ps, 4, 64); ps += __shfl_xor(ps, 8, 64);
        if ((lane & 15) == 0 && m < Co) atomicAdd(pool + (size_t)bc * Co + m, ps);
      } else if (ok) {
        atomicAdd(pool + (size_t)bc * Co + m, ps);
      }
    }
}

// Train-mode statistics of the conv output z = acc + bias for the BatchNorm that follows (models/mn/block_types.py:167-171,
// 177-181 / models/dymn/dy_block.py:313-316, 386-388 under model.train()): per output channel the sum and the sum of squares
// over the block's 256 columns, written as ONE partial per (column tile, channel): part[(tile * 2 + k) * Co + m] - plain
// stores, no atomics, reduced in fp64 by eat_bn_finalize_partials (outer = column tiles, inner = 1).  The standalone
// statistics pass over z (a full read of the conv output) disappears.
//   lane (kq, c): rows (mt0 + i) * 16 + kq * 4 + r, 4 columns -> 16-lane DPP row reduction (the 16 lanes of a kq group hold
//   the 64 columns of the wave), then the 4 waves are combined through LDS (`scratch`: >= 4 * MTW * 16 * 2 floats, free
//   after the k loop - the caller has put a barrier between the loop and this call).
__device__ __forceinline__ float row16_sum_lane15(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));   // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));   // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));   // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));   // row_shr:8
  return v;                                                // lane 15 of every 16-lane row holds the row's sum
}
// YT = bf16: the output was STORED in bf16 - the sums are those of the rounded values (act_io.h)
template <int MTW, typename YT = float>
__device__ __forceinline__ void pw_epilogue_stats(const acc_f32x4 (&acc)[MTW][4], const float* s_bias, float* scratch,
                                                  float* __restrict__ part, int tile, int mt0, int kq, int lane, int wv,
                                                  bool col_ok, int Co) {
  const float ok = col_ok ? 1.0f : 0.0f;
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float bm = s_bias[i * 16 + kq * 4 + r];
      const float v0 = Io<YT>::rnd(acc[i][0][r] + bm), v1 = Io<YT>::rnd(acc[i][1][r] + bm);
      const float v2 = Io<YT>::rnd(acc[i][2][r] + bm), v3 = Io<YT>::rnd(acc[i][3][r] + bm);
      float sm = ((v0 + v1) + (v2 + v3)) * ok;
      float sq = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, v3 * v3))) * ok;
      sm = row16_sum_lane15(sm);
      sq = row16_sum_lane15(sq);
      if ((lane & 15) == 15) {
        float* d = scratch + ((wv * MTW + i) * 16 + kq * 4 + r) * 2;
        d[0] = sm; d[1] = sq;
      }
    }
  __syncthreads();
  for (int e = threadIdx.x; e < MTW * 16 * 2; e += 256) {
    const int row = e >> 1, k = e & 1, m = mt0 * 16 + row;
    if (m < Co) {
      const float t = (scratch[(0 * MTW * 16 + row) * 2 + k] + scratch[(1 * MTW * 16 + row) * 2 + k]) +
                      (scratch[(2 * MTW * 16 + row) * 2 + k] + scratch[(3 * MTW * 16 + row) * 2 + k]);
      part[((size_t)tile * 2 + k) * Co + m] = t;
    }
  }
}

// Backward statistics of the BatchNorm + activation that PRODUCED this conv's input-side tensor, taken in the epilogue of the
// project conv's data-gradient GEMM (round 5): the conv output is dxs = W^T dz_p, the gradient arriving at
// y_d = act(a z_d + b) (models/mn/block_types.py:150-162); the BatchNorm backward of the depthwise conv needs per channel
//   sum g   and   sum g z_d,    g = dxs * act'(a z_d + b)
// (eat_bn_act_bwd_reduce read dxs AND z_d for them: 0.71 ms per mn10 step on the three stem-resolution blocks).  Here dxs is
// still in the accumulators: the epilogue loads the z_d tile, forms g and leaves ONE partial per (column tile, channel) in
// the layout of pw_epilogue_stats - [tile][2][Co] - which eat_bn_bwd_sums_from_tiles reduces in fp64 (sum g z - mean sum g,
// times invstd).  YT = bf16: g is formed from dxs AS STORED.  ZT: storage type of z_d.
// (struct PwGStat { z, a, b, act }: eat_common.h)
// rows of one activation kind (a template parameter: with a run-time kind the compiler evaluated the Hardswish AND the ReLU
// derivative for every element - the epilogue's VALU work, not its loads, made the first version slower than the pass it
// replaces)
template <int MTW, int ACT, typename YT, typename ZT>
__device__ __forceinline__ void pw_gstats_rows(const acc_f32x4 (&acc)[MTW][4], const float* s_bias, const float* s_ab,
                                               float* scratch, const ZT* zb, int mt0, int kq, int lane, int wv, float ok,
                                               int Co, int S) {
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i * 16 + kq * 4 + r;
      int m = mt0 * 16 + row;
      const float rok = m < Co ? ok : 0.0f;
      if (m >= Co) m = Co - 1;
      const float bm = s_bias[row], av = s_ab[row], bv = s_ab[MTW * 16 + row], cv = s_ab[2 * MTW * 16 + row];
      const float4 zv = Io<ZT>::load4(zb + (size_t)m * S);
      const float v0 = Io<YT>::rnd(acc[i][0][r] + bm), v1 = Io<YT>::rnd(acc[i][1][r] + bm);
      const float v2 = Io<YT>::rnd(acc[i][2][r] + bm), v3 = Io<YT>::rnd(acc[i][3][r] + bm);
      float g0, g1, g2, g3;
      if constexpr (ACT == EAT_ACT_RELU) {
        g0 = fmaf(av, zv.x, bv) > 0.0f ? v0 : 0.0f; g1 = fmaf(av, zv.y, bv) > 0.0f ? v1 : 0.0f;
        g2 = fmaf(av, zv.z, bv) > 0.0f ? v2 : 0.0f; g3 = fmaf(av, zv.w, bv) > 0.0f ? v3 : 0.0f;
      } else if constexpr (ACT == EAT_ACT_HSWISH) {
        auto dhs = [](float u) {                       // hardswish'(u) = clamp(u / 3 + 1/2 inside [-3, 3]; 0 below, 1 above)
          const float t = fmaf(u, 1.0f / 3.0f, 0.5f);
          return u < -3.0f ? 0.0f : (u > 3.0f ? 1.0f : t);
        };
        g0 = v0 * dhs(fmaf(av, zv.x, bv)); g1 = v1 * dhs(fmaf(av, zv.y, bv));
        g2 = v2 * dhs(fmaf(av, zv.z, bv)); g3 = v3 * dhs(fmaf(av, zv.w, bv));
      } else {
        g0 = v0; g1 = v1; g2 = v2; g3 = v3;
      }
      float sm = ((g0 + g1) + (g2 + g3)) * rok;
      // sum g (z - c) with c = -b / a, the zero of the pre-activation (within a few sigma of the channel mean): a channel whose
      // |mean| >> sigma would otherwise lose in the fp32 tile partial of sum g z the digits the fp64 finish subtracts
      // (eat_bn_bwd_sums_from_tiles adds (c - mean) sum g back)
      float sq = fmaf(g0, zv.x - cv, fmaf(g1, zv.y - cv, fmaf(g2, zv.z - cv, g3 * (zv.w - cv)))) * rok;
      sm = row16_sum_lane15(sm);
      sq = row16_sum_lane15(sq);
      if ((lane & 15) == 15) {
        float* d = scratch + ((wv * MTW + i) * 16 + kq * 4 + r) * 2;
        d[0] = sm; d[1] = sq;
      }
    }
}

template <int MTW, typename YT = float, typename ZT = float>
__device__ __forceinline__ void pw_epilogue_gstats(const acc_f32x4 (&acc)[MTW][4], const float* s_bias, float* scratch,
                                                   float* __restrict__ part, const PwGStat gs, int tile, int mt0, int kq,
                                                   int lane, int wv, bool col_ok, int bc, int sc_, int Co, int S) {
  // stage a, b and the centring constant c of the block's rows behind the wave partials: scratch[4 MTW 16 2 ...) = [3][MTW 16]
  float* s_ab = scratch + 4 * MTW * 16 * 2;
  for (int e = threadIdx.x; e < 3 * MTW * 16; e += 256) {
    const int k = e / (MTW * 16), row = e - k * (MTW * 16);
    int m = mt0 * 16 + row;
    if (m >= Co) m = Co - 1;
    s_ab[e] = k == 0 ? gs.a[m] : (k == 1 ? gs.b[m] : gstat_center(gs.a[m], gs.b[m]));
  }
  __syncthreads();
  const float ok = col_ok ? 1.0f : 0.0f;
  const ZT* zb = reinterpret_cast<const ZT*>(gs.z) + (size_t)bc * Co * (size_t)S + sc_;
  if (gs.act == EAT_ACT_RELU)                                                       // uniform
    pw_gstats_rows<MTW, EAT_ACT_RELU, YT, ZT>(acc, s_bias, s_ab, scratch, zb, mt0, kq, lane, wv, ok, Co, S);
  else if (gs.act == EAT_ACT_HSWISH)
    pw_gstats_rows<MTW, EAT_ACT_HSWISH, YT, ZT>(acc, s_bias, s_ab, scratch, zb, mt0, kq, lane, wv, ok, Co, S);
  else
    pw_gstats_rows<MTW, EAT_ACT_NONE, YT, ZT>(acc, s_bias, s_ab, scratch, zb, mt0, kq, lane, wv, ok, Co, S);
  __syncthreads();
  for (int e = threadIdx.x; e < MTW * 16 * 2; e += 256) {
    const int row = e >> 1, k = e & 1, m = mt0 * 16 + row;
    if (m < Co) {
      const float t = (scratch[(0 * MTW * 16 + row) * 2 + k] + scratch[(1 * MTW * 16 + row) * 2 + k]) +
                      (scratch[(2 * MTW * 16 + row) * 2 + k] + scratch[(3 * MTW * 16 + row) * 2 + k]);
      part[((size_t)tile * 2 + k) * Co + m] = t;
    }
  }
}

// `act` is wave-uniform: the project layers (no activation) take a branch without any activation math,
// ReLU / Hardswish share the branch-free 4-op form (a third specialisation pushed the 7-8 m-tile kernels
// over 256 VGPRs).
template <int MTW, typename YT = float>
__device__ __forceinline__ void pw_epilogue(const acc_f32x4 (&acc)[MTW][4], const float* s_bias,
                                            const float* __restrict__ res, YT* __restrict__ y,
                                            float* __restrict__ pool, int mt0, int kq, int lane, bool col_ok, int bc,
                                            int sc_, int Co, int S, int act) {
  const ActCoef ac = act_coef(act);
  if (act == EAT_ACT_NONE)
    pw_epilogue_act<MTW, true, YT>(acc, s_bias, res, y, pool, mt0, kq, lane, col_ok, bc, sc_, Co, S, ac);
  else
    pw_epilogue_act<MTW, false, YT>(acc, s_bias, res, y, pool, mt0, kq, lane, col_ok, bc, sc_, Co, S, ac);
}

}  // namespace eat
