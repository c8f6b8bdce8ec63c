// Barrier-free 1x1 convolutions on the bf16 matrix cores (round 2, second half): same arithmetic, packs, lane layout
// and epilogue contract as conv_pw_bf16.hip (bf16x3 split operands or plain bf16, fp32 accumulation), different data
// movement.  conv_pw_bf16.hip stages every 32-row chunk of x through LDS for the 4 waves of a block
// (LDS-DMA -> barrier -> ds_read -> convert -> MFMA -> barrier) although a wave only ever consumes ITS OWN 64 columns,
// and writes the whole (MTW*16) x 256 output tile in one burst at the end; with 192-246 VGPRs that is two blocks per CU
// whose load, compute and store phases overlap only by accident: 3.2-3.7 TB/s on layers that are pure streaming.
// Here every wave is independent (no LDS for x or the weights, no barrier in the main loop, the compiler owns every
// s_waitcnt):
//
//   pw_expand_kernel  (C_in <= 128, no SE scale / residual / pool: the expand convs of models/mn/block_types.py:138-147
//                      and, in training, the data gradient of the project convs): a wave loads its 64 columns of ALL
//                      C_in rows straight into the MFMA B-operand layout (16-byte loads: 4 consecutive columns per
//                      lane = one column of each of 4 n-tiles), converts them ONCE to bf16 hi/lo and keeps them in
//                      registers (<= 128 VGPRs); then it walks down the output rows one 16-row m-tile at a time: A
//                      fragments (16 bytes per lane, L1/L2-resident packed weights) one m-tile ahead, 12 MFMAs per
//                      32-row chunk, bias + activation, 4 x 16-byte stores per lane.  The output - 3-6x the input
//                      bytes on these layers - leaves the CU as a continuous stream under the MFMAs of the next m-tile
//                      instead of a burst after the last k-chunk.
//   pw_kstream_kernel (any C_in, <= 6 m-tiles per block: the project convs, block_types.py:167-171,83,177-181, with SE
//                      scale on the input, residual, pooled sums): a wave owns 64 columns x (MTW*16) rows of output
//                      and streams K in 32-row chunks: chunk c+1 is requested (8 x 16-byte loads per lane, straight
//                      into registers) right after chunk c has been converted, the A fragments of chunk c+1 replace
//                      those of chunk c as the MFMAs consume them.  Epilogue = pw_epilogue.h.
//
// Both are reached through eat_pw_conv_bf16_fwd (conv_pw_bf16.hip) - `EAT_PW_STREAM` selects them (see pw_stream_try).
#include <atomic>
#include <cstdlib>
#include "eat_common.h"
#include "pw_epilogue.h"
#include "bf16_frag.h"

namespace {

using namespace eatfrag;

struct ColGeom {
  bool col_ok;
  int bc, sc_;
};

// the lane's 4 consecutive columns (conv_pw_bf16.hip's mapping: column tile, 64 columns per wave, 4 per lane)
__device__ __forceinline__ ColGeom col_geom(int tile, int wv, int lane, int B, int S) {
  const unsigned N = (unsigned)B * (unsigned)S;
  const unsigned nc = (unsigned)tile * kTileN + 64 * wv + 4 * (lane & 15);
  ColGeom g;
  g.col_ok = nc < N;
  const unsigned ncc = g.col_ok ? nc : N - 4;
  g.bc = (int)(ncc / (unsigned)S);
  g.sc_ = (int)(ncc - (unsigned)g.bc * (unsigned)S);
  return g;
}

// ---------------------------------------------------------------------------------------------------------------
// expand-shaped layers: x resident in registers, output rows streamed
template <int NCH, int NPROD, bool LINEAR>
__global__ __launch_bounds__(256, (NCH <= 2 ? 3 : 2)) void pw_expand_kernel(
    const float* __restrict__ x, const bf16x8* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    int B, int Ci, int Co, int S, int MT, int MTB, int MC, int n_tiles, int act) {
  constexpr int NP2 = NPROD == 3 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float s_bias[1024];   // the block's rows (host: MTB * 16 <= 1024)
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;    // m-chunks of one column tile share an XCD (its L2 holds x)
  const int mchunk = jj % MC, tile = (jj / MC) * 8 + xcd;
  if (tile >= n_tiles) return;
  const int mt0 = mchunk * MTB;
  const int mt1 = (mt0 + MTB) < MT ? (mt0 + MTB) : MT;
  for (int e = tid; e < (mt1 - mt0) * 16; e += 256) {
    const int m = mt0 * 16 + e;
    s_bias[e] = bias[m < Co ? m : Co - 1];
  }
  __syncthreads();                                          // the only barrier of the kernel

  const ColGeom g = col_geom(tile, wv, lane, B, S);
  const int kq = lane >> 4;
  const float* xcol = x + (size_t)g.bc * Ci * S + g.sc_;

  // ---- x -> B fragments of all chunks.  Chunk c+1 is requested before chunk c is converted, so at most two chunks
  // of raw fp32 (64 VGPRs) are in flight next to the fragments built so far.
  bf16x8 bh[NCH][4], bl[NCH][4];
  auto load_rows = [&](int c, float4 (&xr)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = c * kKC + 8 * kq + i;
      xr[i] = *reinterpret_cast<const float4*>(xcol + (size_t)(k < Ci ? k : Ci - 1) * S);   // padded k: finite x, zero w
    }
  };
  {
    float4 xa[8], xb[8];
    load_rows(0, xa);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) {
        if ((c & 1) == 0) load_rows(c + 1, xb); else load_rows(c + 1, xa);
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((c & 1) == 0) split_rows<NPROD>(xa, bh[c], bl[c]); else split_rows<NPROD>(xb, bh[c], bl[c]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- walk down the rows.  Fragment (c, mt, h) of the packed weights sits at byte ((c * MT + mt) * NP2 + h) * 1024 +
  // lane * 16: the lane part is the (loop-invariant) vector offset, the rest a scalar offset.
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (long long)NCH * MT * NP2 * 1024);
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, (long long)B * Co * S * 4);        // host: < 2^31 bytes
  const unsigned a_voff = lane * 16;
  auto frag_soff = [&](int c, int mt, int h) { return (unsigned)(((c * MT + mt) * NP2 + h) * 1024); };
  const eat::ActCoef ac = eat::act_coef(act);
  // row r of the lane's quad in m-tile 0 of its sample; the m-tile adds a scalar offset
  unsigned y_voff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    y_voff[r] = g.col_ok ? (unsigned)((((size_t)g.bc * Co + 4 * kq + r) * S + g.sc_) * 4) : kOOB;
  const int m_full = Co / 16;                               // m-tiles below this have all 16 rows

  // One m-tile: request the NEXT m-tile's fragments and bias first (a whole m-tile of MFMAs ahead of their use),
  // multiply with the current ones, activation, 4 stores.  Two register sets alternate (the loop is unrolled by two) so
  // that nothing is copied at the loop end: a copy would wait for the loads it was meant to hide.
  //   * The accumulators START at the bias (it is the C operand of the first chunk's MFMAs): no add in the epilogue.
  //   * Every store is followed by 8 wait states.  hipcc assumes that a buffer_store_dwordx4 with an SGPR offset has
  //     read its 16 bytes of data when it issues (GCNHazardRecognizer: "this hazard only exists if the instruction is
  //     not using a register in the soffset field") and lets the register allocator's copies for the NEXT row overwrite
  //     the data registers in the very next slot.  On gfx950 the store then picks up the new values in lanes 12-15 of
  //     every 16-lane row - measured: the rows whose registers were re-used right away were wrong in exactly those
  //     lanes, differently from run to run, only under load; with the writers kept away: bit-stable
  //     (tools/dbg_expand.py).  The 4-/8-byte buffer stores of irb.hip / dw_plane.hip are not affected.
  auto m_tile = [&](int mt, const bf16x8 (&cur)[NCH][NP2], bf16x8 (&nxt)[NCH][NP2], const f32x4& bcur, f32x4& bnxt) {
    const int mtn = (mt + 1 < mt1) ? mt + 1 : mt;           // the last one re-reads its own (harmless)
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int h = 0; h < NP2; ++h) nxt[c][h] = buf_load_frag(rw, a_voff, frag_soff(c, mtn, h));
    bnxt = *reinterpret_cast<const f32x4*>(s_bias + (mtn - mt0) * 16 + 4 * kq);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = bcur;               // acc[j][r] = bias of row 4 * kq + r
#pragma unroll
    for (int c = 0; c < NCH; ++c) mfma_chunk<NPROD>(acc, cur[c][0], cur[c][NP2 - 1], bh[c], bl[c]);
    const unsigned y_soff = (unsigned)mt * 16u * (unsigned)S * 4u;
    const bool partial = mt >= m_full;                      // wave-uniform: only the last m-tile of a ragged Co
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v;
      if constexpr (LINEAR)
        v = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      else
        v = f32x4{eat::act_apply(acc[0][r], ac), eat::act_apply(acc[1][r], ac), eat::act_apply(acc[2][r], ac),
                  eat::act_apply(acc[3][r], ac)};
      const unsigned vo = (partial && mt * 16 + 4 * kq + r >= Co) ? kOOB : y_voff[r];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (int)vo, (int)y_soff, 0);
      // the 128-bit data tuple is an INPUT of the wait: nothing can overwrite it before 8 wait states have passed
      asm volatile("s_nop 7" ::"v"(v) : "memory");
    }
  };

  bf16x8 a0[NCH][NP2], a1[NCH][NP2];
  f32x4 b0 = *reinterpret_cast<const f32x4*>(s_bias + 4 * kq), b1;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int h = 0; h < NP2; ++h) a0[c][h] = buf_load_frag(rw, a_voff, frag_soff(c, mt0, h));
  // Enter the loop with nothing pending: the compiler merges the s_waitcnt state of the loop entry with that of the back
  // edge and takes the stricter count - with the first fragments still in flight at the entry it waits for "all but 9"
  // at the loop head of EVERY iteration, which drains the stores of the previous m-tile.
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
  __builtin_amdgcn_sched_barrier(0);
  int mt = mt0;
  for (; mt + 1 < mt1; mt += 2) {
    m_tile(mt, a0, a1, b0, b1);
    m_tile(mt + 1, a1, a0, b1, b0);
  }
  if (mt < mt1) m_tile(mt, a0, a1, b0, b1);
}

// ---------------------------------------------------------------------------------------------------------------
// project-shaped layers: output tile resident in the accumulators, K streamed.
//
// gfx9 returns vector-memory results in issue order on ONE counter (vmcnt), so "wait for load L" means "wait for
// everything issued before L".  The schedule is built around that: per 32-row chunk c
//     wait x(c)  ->  [SE scale]  ->  split into bf16 hi/lo  ->  request x(c+1)  ->  for every m-tile i:
//     12 MFMAs with A(c, i), then request A(c+1, i) into the registers just consumed
// x(c+1) (the HBM stream) is in flight under the whole MFMA phase; the A fragments of chunk c+1 (L2 hits) are younger
// than x(c+1) and are complete when x(c+1) is, one chunk before they are needed - no wait inside the MFMA phase.
template <int MTW, int NPROD, bool SCALE>
__global__ __launch_bounds__(256, (MTW <= 2 ? 3 : 2)) void pw_kstream_kernel(
    const float* __restrict__ x, const bf16x8* __restrict__ wp, const float* __restrict__ bias,
    const float* __restrict__ in_scale, const float* __restrict__ res, float* __restrict__ y, float* __restrict__ pool,
    int B, int Ci, int Co, int S, int MT, int MC, int n_tiles, int act, int ci_x) {
  // ci_x: channels of x.  ci_x == Ci: plain 1x1 conv.  ci_x < Ci: DyMN "K-concat" (conv_pw_bf16.hip): k = bank * ci_x + ci,
  // the scale (B, Ci) carries the attention; ci_x % 32 == 0, so a chunk never straddles two banks.
  constexpr int NP2 = NPROD == 3 ? 2 : 1;
  __shared__ float s_bias[128];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
  const int mchunk = jj % MC, tile = (jj / MC) * 8 + xcd;
  if (tile >= n_tiles) return;
  const int mt0 = mchunk * MTW;
  if (tid < 128) {
    const int m = mt0 * 16 + tid;
    s_bias[tid] = bias[m < Co ? m : Co - 1];
  }
  __syncthreads();                                          // the only barrier of the kernel

  const ColGeom g = col_geom(tile, wv, lane, B, S);
  const int kq = lane >> 4;
  const float* xcol = x + (size_t)g.bc * ci_x * S + g.sc_;
  const float* scp = SCALE ? in_scale + (size_t)g.bc * Ci : nullptr;
  const int n_chunks = (Ci + kKC - 1) / kKC;

  float4 xr[8];
  float4 sc[2];
  auto load_rows = [&](int c) {
    if constexpr (SCALE) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {                          // Ci % 4 == 0: a group of 4 k is inside or outside as a whole
        const int k = c * kKC + 8 * kq + 4 * q;
        sc[q] = *reinterpret_cast<const float4*>(scp + (k < Ci ? k : Ci - 4));
      }
    }
    const int kx0 = (c * kKC) % ci_x;                       // row of x the chunk starts at (wave-uniform)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = c * kKC + 8 * kq + i;
      const int kx = kx0 + 8 * kq + i;
      xr[i] = *reinterpret_cast<const float4*>(xcol + (size_t)(k < Ci ? kx : ci_x - 1) * S);   // padded k: finite x, zero w
    }
  };
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wp, (long long)n_chunks * MT * NP2 * 1024);
  const unsigned a_voff = lane * 16;
  bf16x8 af[MTW][NP2];
  auto load_frag = [&](int c, int i) {
    int mt = mt0 + i;
    if (mt >= MT) mt = MT - 1;                               // rows >= Co are never stored
#pragma unroll
    for (int h = 0; h < NP2; ++h) af[i][h] = buf_load_frag(rw, a_voff, (unsigned)(((c * MT + mt) * NP2 + h) * 1024));
  };

  f32x4 acc[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < MTW; ++i) load_frag(0, i);
  load_rows(0);
  // enter the loop with nothing pending (see pw_expand_kernel: the loop-entry state must not be stricter than the back edge)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
  __builtin_amdgcn_sched_barrier(0);

  for (int c = 0; c < n_chunks; ++c) {
    const int cn = (c + 1 < n_chunks) ? c + 1 : c;          // the last chunk re-requests itself (harmless, branch-free)
    if constexpr (SCALE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float sv = i < 4 ? sc[0][i & 3] : sc[1][i & 3];
        xr[i].x *= sv; xr[i].y *= sv; xr[i].z *= sv; xr[i].w *= sv;
      }
    }
    bf16x8 bh[4], bl[4];
    split_rows<NPROD>(xr, bh, bl);
    load_rows(cn);                                           // in flight under this chunk's MFMAs
    __builtin_amdgcn_sched_barrier(0);                       // (hipcc sinks these loads below 3/4 of the MFMAs otherwise)
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      mfma_chunk<NPROD>(acc[i], af[i][0], af[i][NP2 - 1], bh, bl);
      load_frag(cn, i);                                      // into the registers the MFMAs above have just read
    }
  }

  eat::pw_epilogue<MTW>(acc, s_bias, res, y, pool, mt0, kq, lane, g.col_ok, g.bc, g.sc_, Co, S, act);
}

template <int NCH, int NPROD>
int launch_expand(hipStream_t s, const float* x, const void* wp, const float* bias, float* y, int B, int Ci, int Co, int S,
                  int act) {
  const int MT = (Co + 15) / 16;
  const int n_tiles = (int)(((long long)B * S + kTileN - 1) / kTileN);
  // all rows in one block when the column tiles alone fill the chip twice; otherwise split the rows (x is re-read
  // from the XCD's L2 by the other m-chunks)
  int MC = 1;
  while (n_tiles * MC < 448 && (MT + 2 * MC - 1) / (2 * MC) >= 4) MC *= 2;
  while ((MT + MC - 1) / MC > 64) ++MC;                      // s_bias holds 64 m-tiles
  const int MTB = (MT + MC - 1) / MC;
  MC = (MT + MTB - 1) / MTB;
  const int tiles8 = (n_tiles + 7) / 8 * 8;
  auto kern = act == EAT_ACT_NONE ? pw_expand_kernel<NCH, NPROD, true> : pw_expand_kernel<NCH, NPROD, false>;
  hipLaunchKernelGGL(kern, dim3(tiles8 * MC), dim3(256), 0, s, x, reinterpret_cast<const bf16x8*>(wp), bias, y, B, Ci,
                     Co, S, MT, MTB, MC, n_tiles, act);
  return eat::check_launch("eat_pw_conv_bf16_fwd (expand)");
}

template <int MTW, int NPROD>
int launch_kstream(hipStream_t s, const float* x, const void* wp, const float* bias, const float* in_scale,
                   const float* res, float* y, float* pool, int B, int Ci, int Co, int S, int MT, int act, int ci_x) {
  const int MC = (MT + MTW - 1) / MTW;
  const int n_tiles = (int)(((long long)B * S + kTileN - 1) / kTileN);
  const int tiles8 = (n_tiles + 7) / 8 * 8;
  auto kern = in_scale ? pw_kstream_kernel<MTW, NPROD, true> : pw_kstream_kernel<MTW, NPROD, false>;
  hipLaunchKernelGGL(kern, dim3(tiles8 * MC), dim3(256), 0, s, x, reinterpret_cast<const bf16x8*>(wp), bias, in_scale,
                     res, y, pool, B, Ci, Co, S, MT, MC, n_tiles, act, ci_x);
  return eat::check_launch("eat_pw_conv_bf16_fwd (kstream)");
}

template <int NPROD>
int try_stream(hipStream_t s, const float* x, const void* wp, const float* bias, const float* in_scale, const float* res,
               float* y, float* pool, int B, int Ci, int Co, int S, int act, int mode, int ci_x) {
  const int n_chunks = (Ci + kKC - 1) / kKC;
  if ((mode & 1) && ci_x == Ci && !in_scale && !res && !pool && y && n_chunks <= 4 && Co >= 2 * Ci &&
      (long long)B * Co * S * 4 < 0x7fffffffLL) {
    switch (n_chunks) {
      case 1: return launch_expand<1, NPROD>(s, x, wp, bias, y, B, Ci, Co, S, act);
      case 2: return launch_expand<2, NPROD>(s, x, wp, bias, y, B, Ci, Co, S, act);
      case 3: return launch_expand<3, NPROD>(s, x, wp, bias, y, B, Ci, Co, S, act);
      default: return launch_expand<4, NPROD>(s, x, wp, bias, y, B, Ci, Co, S, act);
    }
  }
  // bit 1 (the default): project-shaped layers where the K-streaming kernel measured faster in the mn10 forward at
  // B = 256 - long reductions (C_in >= 160) that need no more row chunks than the LDS-staged kernel (<= 6 m-tiles, or as
  // many chunks of <= 6 as of <= 8): 240->80 36->28 us, 672->160 41->28, 960->160 59->38; NOT 120->40 @ 16x125 (4
  // chunks of K: 82 vs 86 us) and NOT 7 m-tiles (4 + 3 against one block of 7: 480->112 76 vs 87 us).
  // bit 2: every other layer as row chunks of the K-streaming kernel (A/B and tests).
  const int MT = (Co + 15) / 16;
  const bool measured_win = Co <= 2 * Ci && Ci >= 160 && (MT + 5) / 6 == (MT + 7) / 8;
  if (ci_x != Ci ? (mode & 8) != 0 : (((mode & 2) && measured_win) || (mode & 4))) {
    const int MC = (MT + 5) / 6;                             // at most 6 m-tiles of accumulators + fragments per wave
    const int mtw = (MT + MC - 1) / MC;
#define EAT_CASE(n) \
  case n: return launch_kstream<n, NPROD>(s, x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, MT, act, ci_x);
    switch (mtw) {
      EAT_CASE(1) EAT_CASE(2) EAT_CASE(3) EAT_CASE(4) EAT_CASE(5) EAT_CASE(6)
      default: break;
    }
#undef EAT_CASE
  }
  return 1;
}

}  // namespace

namespace {
std::atomic<int>& stream_mode() {
  static std::atomic<int> mode{getenv("EAT_PW_STREAM") ? atoi(getenv("EAT_PW_STREAM")) & 15 : 2};
  return mode;
}
}  // namespace

extern "C" int eat_pw_stream_mode(int mode) {
  return mode >= 0 ? stream_mode().exchange(mode & 15) : stream_mode().load();
}

namespace eat {

// Returns 1 when the shape (or the EAT_PW_STREAM switch: bit 0 = expand kernel, bit 1 = k-stream kernel for project-shaped
// layers, bit 2 = k-stream kernel for whatever is left, bit 3 = k-stream kernel for DyMN's K-concat launches) leaves the
// layer to the LDS-staged kernel of conv_pw_bf16.hip; otherwise the launch status.  Caller has checked S % 4 == 0,
// Ci % 4 == 0 (and ci_x % 32 == 0, in_scale != NULL when ci_x != Ci).
int pw_stream_try(const float* x, const void* wp, const float* bias, const float* in_scale, const float* res, float* y,
                  float* pool, int B, int Ci, int Co, int S, int act, int split, int ci_x, hipStream_t s) {
  const int mode = stream_mode().load(std::memory_order_relaxed);
  if (!mode) return 1;
  if ((long long)B * S > 0x7fff0000LL) return 1;
  return split ? try_stream<3>(s, x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, act, mode, ci_x)
               : try_stream<1>(s, x, wp, bias, in_scale, res, y, pool, B, Ci, Co, S, act, mode, ci_x);
}

}  // namespace eat
