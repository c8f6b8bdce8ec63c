// Register-resident inverted-residual block for gfx950 (round 2 replacement of the LDS-staged mbconv kernel for the
// early, bandwidth-bound blocks):
//   expand 1x1 + BN + act  ->  depthwise k x k + BN + act  ->  project 1x1 + BN (+ residual)        [PROJ]
//   expand 1x1 + BN + act  ->  depthwise k x k + BN + act  (+ SE squeeze sums), output to HBM          [!PROJ]
//   reference: InvertedResidual.forward (models/mn/block_types.py:138-181), SE mean (:72-73).
//
// The round-1 kernel staged the expanded patch, the depthwise output and the project input through LDS with 3-4
// workgroup barriers per 16 channels; counters: MFMA pipe 27 % busy, 39 % of wave cycles parked in waits, 37 % issue
// stalls, 9 of 16 wave slots idle in the expand phase.  Here one WAVE owns a strip of 32 output columns and marches
// down the rows; nothing but the weights ever touches LDS and there is no barrier after the prologue:
//
//   * expand: v_mfma_f32_16x16x4_f32 with N = 16 consecutive (stride 1) or every-other (stride 2: an "even" tile
//     E(u) = column 2 oc and an "odd" tile O(u) = column 2 oc - 1) input columns of one row; the B operand is loaded
//     straight from global memory (lane (k, n) = channel 4 ks + k, column n: 64-byte segments), the A operand from LDS;
//     the accumulator starts at the BN bias;
//   * the C layout of the result (lane (kq, n) holds channels 4 kq + r, r = 0..3, at column n) IS the layout the
//     depthwise conv wants: vertical taps are other registers of the same lane (a window of the last K - S expanded
//     rows stays in registers while the wave marches down), horizontal taps are neighbouring lanes of the same
//     16-lane row, i.e. DPP row shifts folded into the FMAs (lane 0 / 15 take the neighbour tile's edge lane through
//     a second row_shl:15 / row_shr:15 FMA).  The outermost column(s) of a strip have no neighbour: 30 or 31 of the 32
//     columns are valid outputs, strips overlap by the halo;
//   * the same registers are the B operand of the project MFMA: k-step r of chunk c multiplies channels
//     {16 c + 4 k + r}, so the project A fragments are re-ordered once (prologue) instead of moving any data;
//   * positions outside the image are forced to 0 after the activation (the depthwise conv zero-pads the ACTIVATED map).
//
// A wave's serial work is one row at a time: S new expanded rows (all chunks), the depthwise row, the project row.
// The next row's B operands are requested right after the last chunk's expand MFMAs have issued, i.e. one depthwise +
// project phase ahead of their use.
#include <cstdlib>
#include "eat_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kWaves = 4;       // waves per block (they share the LDS weights only)

struct IrbArgs {
  const float *x, *wpe, *bias_e, *wd, *bias_d, *wpp, *bias_p, *res;
  float *y, *pool;
  int B, Cin, Cexp, Cout, F, T, Fo, To;
  int Fm, Tm;                             // FRONT: log-mel plane (F, T are the stem plane)
  int MT;                                 // 16-channel chunks of the expanded tensor
  int n_strips, n_parts, rows_per_part;   // work decomposition: item = (sample, column strip, row range)
  int n_items;
};

// Raw buffer access (128-bit descriptor in SGPRs + 32-bit per-lane byte offset + scalar byte offset): the per-lane
// offsets are a handful of loop-invariant registers and everything that changes per row / k-step / channel is scalar.
// With flat pointers hipcc materialised one 64-bit VGPR address per load and hoisted dozens of them out of the loops
// (256 VGPRs + spills).  A lane whose offset is kOOB (>= num_records) loads 0 / stores nothing (hardware range check).
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

constexpr int kShr1 = 0x111, kShl1 = 0x101, kShl15 = 0x10F, kShr15 = 0x11F;   // DPP row shifts
template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {      // lane j <- lane j -/+ n of its 16-lane row; 0 where there is none
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Depthwise taps, factored so that the horizontal shifts are applied ONCE per output instead of once per tap: per input
// row only plain FMAs into three partial sums - L (taps whose input is the left neighbour u - 1), M (same lane), R (right
// neighbour u + 1) - and after the last row  d(u) = M(u) + L(u - 1) + R(u + 1)  with two DPP row shifts (+ the neighbour
// tile's edge lane for lanes 0 / 15).  Shifting is linear, so this equals shifting every input (zero fill included).
template <int NT>
struct DwAcc {
  f32x4 l[NT], m[NT], r[NT];
};
template <int NT>
__device__ __forceinline__ void fma_tiles(f32x4 (&d)[NT], const f32x4* row, const f32x4 w) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) d[t][r] = fmaf(w[r], row[t][r], d[t][r]);
}
// one input row's contribution; row = [E tiles (NT)] [O tiles (NT), stride 2 only]
template <int K, int S, int NT>
__device__ __forceinline__ void dw_row(DwAcc<NT>& d, const f32x4* row, const f32x4* w) {
  constexpr int kNT = NT;
  if constexpr (S == 1 && K == 3) {                 // columns oc-1, oc, oc+1
    fma_tiles(d.l, row, w[0]); fma_tiles(d.m, row, w[1]); fma_tiles(d.r, row, w[2]);
  } else if constexpr (S == 2 && K == 3) {          // columns 2oc-1, 2oc, 2oc+1 = O(u), E(u), O(u+1)
    fma_tiles(d.m, row + kNT, w[0]); fma_tiles(d.m, row, w[1]); fma_tiles(d.r, row + kNT, w[2]);
  } else {                                          // S == 2, K == 5: E(u-1), O(u), E(u), O(u+1), E(u+1)
    fma_tiles(d.l, row, w[0]); fma_tiles(d.m, row + kNT, w[1]); fma_tiles(d.m, row, w[2]);
    fma_tiles(d.r, row + kNT, w[3]); fma_tiles(d.r, row, w[4]);
  }
}
template <int K, int S, int NT>
__device__ __forceinline__ void dw_finish(const DwAcc<NT>& a, f32x4 (&d)[NT]) {
  constexpr int kNT = NT;
#pragma unroll
  for (int t = 0; t < kNT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = a.m[t][r] + dpp0<kShl1>(a.r[t][r]);
      if (t + 1 < kNT) v += dpp0<kShr15>(a.r[t + 1][r]);
      if constexpr (!(S == 2 && K == 3)) {
        v += dpp0<kShr1>(a.l[t][r]);
        if (t > 0) v += dpp0<kShl15>(a.l[t - 1][r]);
      }
      d[t][r] = v;
    }
}

// K, S: depthwise kernel / stride; NKS = Cin / 4; NT = output column tiles (of 16) per wave; MTI = 16-channel chunks
// marched together (all of them with PROJ; without, ceil(MT / MTI) marches re-read the input); MTO = project m-tiles;
// PF: the next row's B operands are requested one whole row ahead into a second register set (otherwise after the last
// chunk's expand MFMAs, into the same registers)
// FRONT: the network front (stem 3x3 / stride 2 conv of the 1-channel log-mel + BN + Hardswish -> first block: depthwise
//   3x3 + BN + act -> project 1x1 + BN + residual; models/mn/model.py:124-133, block_types.py:150-181).  The stem IS the
//   "expand" of this kernel: an MFMA with K = 9 taps (padded to 12, NKS = 3) whose B operand is the im2col gather of the
//   log-mel (lane (k, n): tap 4 ks + k of stem column n, zero outside the image through the buffer range check); the
//   residual (= the stem output) is the middle row of the register window, in the accumulator layout already.
//   F, T (a.F, a.T) are then the STEM plane, a.Fm / a.Tm the log-mel plane; ACT_E = Hardswish.
template <int K, int S, int NKS, int NT, int MTI, int MTO, int ACT, bool PROJ, bool PF, int ACT_E = ACT, bool FRONT = false>
__global__ __launch_bounds__(64 * kWaves, ((NT == 1 && MTI * K < 25 && K != 5) ? 3 : 2)) void irb_kernel(const IrbArgs a) {
  static_assert(!FRONT || (K == 3 && S == 1 && NKS == 3 && MTI == 1 && PROJ), "front = stem + 3x3/s1 block with project");
  constexpr int kNT = NT;
  constexpr int P_ = (K - 1) / 2, KK = K * K;
  constexpr int KW = K - S;                 // expanded rows kept between output rows
  constexpr int TI = kNT * S;               // tiles per input row (stride 2: even + odd)
  constexpr int ULO = (K == 3 && S == 2) ? 0 : 1, UHI = 16 * kNT - 2;   // lanes (tile-major index u) with a valid output
  constexpr int VO = UHI - ULO + 1;
  static_assert((K == 3 && (S == 1 || S == 2)) || (K == 5 && S == 2), "unsupported depthwise geometry");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int MT = a.MT;
  float* As = smem;                          // [NKS][MT][64]       expand A fragments (the eat_pw_prepack layout)
  float* Wd = As + NKS * MT * 64;            // [MT][KK][16]        depthwise taps, channel-minor
  float* Be = Wd + MT * KK * 16;             // [MT*16]             expand bias
  float* Bd = Be + MT * 16;                  // [MT*16]             depthwise bias
  float* Ap = Bd + MT * 16;                  // [MT*4][MTO][64]     project A fragments, k order {16c + 4k + r}
  float* Bp = Ap + (PROJ ? MT * 4 * MTO * 64 : 0);   // [MTO*16]
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
  // the wave index is wave-uniform, but only readfirstlane tells the compiler: with it the work-item geometry (sample,
  // strip, rows), every row address and every loop / image-border branch live in SGPRs
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cexp = a.Cexp, F = a.F, T = a.T, Fo = a.Fo, To = a.To;

  // ---- prologue: weights -> LDS (zero beyond Cexp / Cout)
  for (int i = tid; i < NKS * MT * 64; i += 64 * kWaves) {
    if constexpr (FRONT) {                   // stem weights (C, 9) -> A fragments over the 12 (9 + 3 zero) taps
      const int l = i & 63, c = (i >> 6) % MT, ks = (i >> 6) / MT;
      const int ch = c * 16 + (l & 15), tap = 4 * ks + (l >> 4);
      As[i] = (ch < Cexp && tap < 9) ? a.wpe[ch * 9 + tap] : 0.0f;
    } else {
      As[i] = a.wpe[i];
    }
  }
  for (int i = tid; i < MT * KK * 16; i += 64 * kWaves) {
    const int ch = (i / (KK * 16)) * 16 + (i & 15), tap = (i >> 4) % KK;
    Wd[i] = ch < Cexp ? a.wd[(size_t)ch * KK + tap] : 0.0f;
  }
  for (int i = tid; i < MT * 16; i += 64 * kWaves) {
    Be[i] = i < Cexp ? a.bias_e[i] : 0.0f;
    Bd[i] = i < Cexp ? a.bias_d[i] : 0.0f;
  }
  if constexpr (PROJ) {
    for (int i = tid; i < MT * 4 * MTO * 64; i += 64 * kWaves) {
      const int l = i & 63, mo = (i >> 6) % MTO, cr = (i >> 6) / MTO, c = cr >> 2, r = cr & 3;
      const int ks = 4 * c + (l >> 4);                    // source k-step holds channels 4 ks .. 4 ks + 3
      Ap[i] = 4 * ks < Cexp ? a.wpp[((size_t)ks * MTO + mo) * 64 + r * 16 + (l & 15)] : 0.0f;
    }
    for (int i = tid; i < MTO * 16; i += 64 * kWaves) Bp[i] = i < a.Cout ? a.bias_p[i] : 0.0f;
  }
  __syncthreads();

  const int item = blockIdx.x * kWaves + wv;
  if (item >= a.n_items) return;             // no barrier below
  const int per_b = a.n_strips * a.n_parts;
  const int b = item / per_b, rem = item - b * per_b;
  const int strip = rem / a.n_parts, part = rem - strip * a.n_parts;
  const int o0 = strip * VO;
  const int i0 = part * a.rows_per_part;
  const int i1 = (i0 + a.rows_per_part) < Fo ? (i0 + a.rows_per_part) : Fo;
  if (i0 >= i1) return;

  // ---- per-lane column geometry (constant while marching down)
  const int plane = F * T, plane_o = Fo * To;
  unsigned lb[TI];                           // byte offset of (channel kq, clamped column) inside the sample
  float cmask[TI];                           // 1 inside the image, 0 outside
  unsigned ob[kNT];                          // byte offset of (channel kq*4, output column) inside the output sample
  bool ov[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    const int u = 16 * t + n;
    const int oc = o0 + u - ULO;
    ov[t] = u >= ULO && u <= UHI && oc < To;
    ob[t] = 4u * (unsigned)(kq * 4 * plane_o + oc);
#pragma unroll
    for (int h = 0; h < S; ++h) {            // h = 0: even tile (or the only one), h = 1: odd tile
      const int ci = S * oc - h;
      const bool in = ci >= 0 && ci < T;
      cmask[h * kNT + t] = in ? 1.0f : 0.0f;
      lb[h * kNT + t] = 4u * (unsigned)(kq * plane + (ci < 0 ? 0 : (ci >= T ? T - 1 : ci)));
    }
  }
  unsigned fv[FRONT ? NKS : 1][TI];          // FRONT: byte offset of tap 4 ks + kq (row di, column 2 st - 1 + dj) or kOOB
  int fdi[FRONT ? NKS : 1];
  if constexpr (FRONT) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int tap = 4 * ks + kq, di = tap / 3, dj = tap - 3 * di;
      fdi[ks] = di;
#pragma unroll
      for (int t = 0; t < kNT; ++t) {
        const int st = o0 + 16 * t + n - ULO, cm = 2 * st - 1 + dj;
        fv[ks][t] = (tap < 9 && cm >= 0 && cm < a.Tm) ? 4u * (unsigned)(di * a.Tm + cm) : kOOB;
      }
    }
  }
  const int Cy = PROJ ? a.Cout : Cexp;       // channels of y (and of the residual)
  const __amdgpu_buffer_rsrc_t xr_ = FRONT ? make_rsrc(a.x + (size_t)b * a.Fm * a.Tm, 4u * (unsigned)(a.Fm * a.Tm))
                                           : make_rsrc(a.x + (size_t)b * a.Cin * plane, 4u * (unsigned)(a.Cin * plane));
  const __amdgpu_buffer_rsrc_t yr_ = make_rsrc(a.y + (size_t)b * Cy * plane_o, 4u * (unsigned)(Cy * plane_o));
  const __amdgpu_buffer_rsrc_t rr_ = make_rsrc(a.res ? a.res + (size_t)b * Cy * plane_o : a.y, a.res ? 4u * (unsigned)(Cy * plane_o) : 0u);

  float xb[S][NKS][TI];                      // B operands of the S new input rows of this output row
  float xn[PF ? S : 1][PF ? NKS : 1][PF ? TI : 1];   // ... and of the next output row (PF)
  auto load_into = [&](int i, auto& dst) {   // input rows S i - P + KW + j, j < S
    if constexpr (FRONT) {
      const int sf = i - P_ + KW;              // stem row (the "input row" of the depthwise conv)
      if (sf < 0 || sf >= F) return;           // zeroed in expand(); wave-uniform
      const int fb = 2 * sf - 1;               // first log-mel row under the stem tap window
      if (fb >= 0 && fb + 2 < a.Fm) {
        const unsigned ro = 4u * (unsigned)fb * (unsigned)a.Tm;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) dst[0][ks][ti] = buf_load(xr_, fv[ks][ti], ro);
      } else {                                 // top / bottom edge of the log-mel: rows outside read 0 (conv zero padding)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int fr = fb + fdi[ks];
          const bool rin = fr >= 0 && fr < a.Fm;
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) {
            const unsigned v = (rin && fv[ks][ti] != kOOB) ? fv[ks][ti] - 4u * (unsigned)(fdi[ks] * a.Tm) + 4u * (unsigned)(fr * a.Tm) : kOOB;
            dst[0][ks][ti] = buf_load(xr_, v, 0u);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      int fi = S * i - P_ + KW + j;
      fi = fi < 0 ? 0 : (fi >= F ? F - 1 : fi);          // out-of-image rows are zeroed in expand()
      const unsigned ro = 4u * (unsigned)fi * (unsigned)T;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) dst[j][ks][ti] = buf_load(xr_, lb[ti], ro + 16u * (unsigned)ks * (unsigned)plane);
    }
  };
  auto load_rows = [&](int i) { load_into(i, xb); };
  // expanded row of chunk c from B operands xr; fi = its input row (wave-uniform)
  auto expand = [&](int c, const float (&xr)[NKS][TI], int fi, f32x4 (&e)[TI]) {
    if (fi < 0 || fi >= F) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) e[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    const f32x4 be = *reinterpret_cast<const f32x4*>(Be + c * 16 + kq * 4);
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) e[ti] = be;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const float av = As[(ks * MT + c) * 64 + lane];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) e[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xr[ks][ti], e[ti], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) e[ti][r] = eat::activate<ACT_E>(e[ti][r]) * cmask[ti];
  };

  const int n_groups = (MT + MTI - 1) / MTI;  // PROJ: 1 (all chunks marched together); otherwise MTI chunks per march
  for (int g = 0; g < n_groups; ++g) {
    const int c0 = g * MTI;
    f32x4 win[MTI][KW][TI];
    float psum[MTI][4];
#pragma unroll
    for (int c = 0; c < MTI; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) psum[c][r] = 0.0f;

    // ---- prime the window: input rows S i0 - P + w, w < KW (loaded through the same S-row loader)
    {
      // rows come in groups of S: group q covers window rows q S .. q S + S - 1  (KW = 2, S = 1: two groups; KW = 1,
      // S = 2: half a group; KW = 3, S = 2: one and a half) - load "output row" i0 - ceil(KW / S) + q and keep what falls
      // inside the window
      constexpr int NG = (KW + S - 1) / S;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const int iq = i0 - NG + q;          // pseudo output row whose new rows are S iq - P + KW + j
        load_rows(iq);
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const int w = KW - NG * S + q * S + j;           // window slot of that row (may be < 0: not needed)
          if (w >= 0) {
            const int fi = S * iq - P_ + KW + j;
#pragma unroll
            for (int c = 0; c < MTI; ++c) expand(c0 + c < MT ? c0 + c : MT - 1, xb[j], fi, win[c][w]);
          }
        }
      }
    }
    load_rows(i0);

    for (int i = i0; i < i1; ++i) {
      if constexpr (PF) {
        if (i + 1 < i1) load_into(i + 1, xn);            // a whole row of arithmetic ahead of its use
      }
      f32x4 accp[PROJ ? MTO : 1][kNT];
      f32x4 rres[PROJ ? MTO : 1][kNT];
      if constexpr (PROJ) {
#pragma unroll
        for (int mo = 0; mo < MTO; ++mo)
#pragma unroll
          for (int t = 0; t < kNT; ++t) {
            accp[mo][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            // residual of this output row: requested before the row's arithmetic (all zeros without a residual: the
            // descriptor then has 0 records)
            if constexpr (FRONT) {
              rres[mo][t] = win[0][KW - 1][t];           // the stem output of this row: already in the accumulator layout
            } else {
              const unsigned vo = (ov[t] && mo * 16 + kq * 4 < a.Cout) ? ob[t] : kOOB;
#pragma unroll
              for (int r = 0; r < 4; ++r)
                rres[mo][t][r] = buf_load(rr_, vo, 4u * (unsigned)((mo * 16 + r) * plane_o + i * To));
            }
          }
      }
      // (a software-pipelined variant - the expand MFMAs of chunk c + 1 issued before the depthwise arithmetic of chunk c -
      //  measured slower on every block and was removed in round 4)
#pragma unroll
      for (int c = 0; c < MTI; ++c) {
        const bool c_ok = c0 + c < MT;       // a partial last group re-does chunk MT - 1 with its stores masked
        const int cg = c_ok ? c0 + c : MT - 1;
        // compiler barrier: the weights in LDS are loop-invariant, and hipcc would otherwise hoist ALL of them (~250
        // registers for 5 chunks) out of the row loop and spill; they are re-read per chunk instead (LDS has the bandwidth)
        asm volatile("" ::: "memory");
        f32x4 enew[S][TI];
#pragma unroll
        for (int j = 0; j < S; ++j) expand(cg, xb[j], S * i - P_ + KW + j, enew[j]);
        if constexpr (!PF) {
          if (c == MTI - 1 && i + 1 < i1) load_rows(i + 1);   // the B operands are free: next row's loads fly from here
        }
        // depthwise row: K input rows = the window (KW) + the new rows (S)
        const f32x4 bd = *reinterpret_cast<const f32x4*>(Bd + cg * 16 + kq * 4);
        DwAcc<NT> da;
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          da.m[t] = bd;
          da.l[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          da.r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int di = 0; di < K; ++di) {
          asm volatile("" ::: "memory");
          f32x4 w[K];
#pragma unroll
          for (int dj = 0; dj < K; ++dj) w[dj] = *reinterpret_cast<const f32x4*>(Wd + (cg * KK + di * K + dj) * 16 + kq * 4);
          if (di < KW) dw_row<K, S, NT>(da, win[c][di], w);
          else dw_row<K, S, NT>(da, enew[di - KW], w);
        }
        f32x4 d[kNT];
        dw_finish<K, S, NT>(da, d);
#pragma unroll
        for (int t = 0; t < kNT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) d[t][r] = eat::activate<ACT>(d[t][r]);
        if constexpr (PROJ) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mo = 0; mo < MTO; ++mo) {
              const float ap = Ap[((cg * 4 + r) * MTO + mo) * 64 + lane];
#pragma unroll
              for (int t = 0; t < kNT; ++t) accp[mo][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, d[t][r], accp[mo][t], 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int t = 0; t < kNT; ++t) {
            const bool ok = ov[t] && c_ok && cg * 16 + kq * 4 < Cexp;  // channel counts are multiples of 8
            const unsigned vo = ok ? ob[t] : kOOB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              buf_store(d[t][r], yr_, vo, 4u * (unsigned)((cg * 16 + r) * plane_o + i * To));
              psum[c][r] += ok ? d[t][r] : 0.0f;
            }
          }
        }
        // slide the window by S rows
#pragma unroll
        for (int w = 0; w + S < KW; ++w)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) win[c][w][ti] = win[c][w + S][ti];
#pragma unroll
        for (int j = 0; j < S; ++j)
          if (KW - S + j >= 0) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) win[c][KW - S + j][ti] = enew[j][ti];
          }
      }
      if constexpr (PF) {
#pragma unroll
        for (int j = 0; j < S; ++j)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) xb[j][ks][ti] = xn[j][ks][ti];
      }
      if constexpr (PROJ) {
#pragma unroll
        for (int mo = 0; mo < MTO; ++mo) {
          const f32x4 bp = *reinterpret_cast<const f32x4*>(Bp + mo * 16 + kq * 4);
#pragma unroll
          for (int t = 0; t < kNT; ++t) {
            const unsigned vo = (ov[t] && mo * 16 + kq * 4 < a.Cout) ? ob[t] : kOOB;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              buf_store(accp[mo][t][r] + bp[r] + rres[mo][t][r], yr_, vo, 4u * (unsigned)((mo * 16 + r) * plane_o + i * To));
          }
        }
      }
    }
    if constexpr (!PROJ) {
      if (a.pool) {                          // SE squeeze: plane sums of this strip's rows, one atomic per channel
#pragma unroll
        for (int c = 0; c < MTI; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float s = psum[c][r];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            const int ch = (c0 + c) * 16 + kq * 4 + r;
            if (n == 0 && c0 + c < MT && ch < Cexp) atomicAdd(a.pool + (size_t)b * Cexp + ch, s);
          }
      }
    }
  }
}

template <int K, int S, int NKS, int NT, int MTI, int MTO, int ACT, bool PROJ, bool PF>
int launch_irb(IrbArgs a, hipStream_t s) {
  constexpr int ULO = (K == 3 && S == 2) ? 0 : 1, UHI = 16 * NT - 2, VO = UHI - ULO + 1;
  a.MT = (a.Cexp + 15) / 16;
  a.n_strips = (a.To + VO - 1) / VO;
  // enough waves to fill the chip a few times over, but row ranges of >= 8 rows (each range re-expands K - S halo rows)
  constexpr int target = 6144;                               // work items per launch (measured: 3072 ... 12288 within 1 %)
  const long long strips = (long long)a.B * a.n_strips;
  int parts = (int)((target + strips - 1) / strips);
  const int max_parts = a.Fo / 8 > 1 ? a.Fo / 8 : 1;
  parts = parts < 1 ? 1 : (parts > max_parts ? max_parts : parts);
  a.rows_per_part = (a.Fo + parts - 1) / parts;
  a.n_parts = (a.Fo + a.rows_per_part - 1) / a.rows_per_part;
  const long long items = strips * a.n_parts;
  if (items > 0x7fffffffLL) return eat::fail(EAT_EINVAL, "eat_mbconv_fwd: too many work items");
  a.n_items = (int)items;
  const size_t smem = sizeof(float) * ((size_t)NKS * a.MT * 64 + (size_t)a.MT * K * K * 16 + 2 * (size_t)a.MT * 16 +
                                       (PROJ ? (size_t)a.MT * 4 * MTO * 64 + MTO * 16 : 0));
  auto kern = irb_kernel<K, S, NKS, NT, MTI, MTO, ACT, PROJ, PF, ACT, false>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eat::fail(EAT_ELAUNCH, "irb: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((items + kWaves - 1) / kWaves)), dim3(64 * kWaves), smem, s, a);
  return eat::check_launch("irb_kernel");
}

template <int NT, int ACT, bool PF>
int launch_front(IrbArgs a, hipStream_t s) {
  constexpr int VO = 16 * NT - 2;
  a.MT = 1;
  a.n_strips = (a.To + VO - 1) / VO;
  constexpr int target = 6144;
  const long long strips = (long long)a.B * a.n_strips;
  int parts = (int)((target + strips - 1) / strips);
  const int max_parts = a.Fo / 8 > 1 ? a.Fo / 8 : 1;
  parts = parts < 1 ? 1 : (parts > max_parts ? max_parts : parts);
  a.rows_per_part = (a.Fo + parts - 1) / parts;
  a.n_parts = (a.Fo + a.rows_per_part - 1) / a.rows_per_part;
  const long long items = strips * a.n_parts;
  if (items > 0x7fffffffLL) return eat::fail(EAT_EINVAL, "eat_front_fwd: too many work items");
  a.n_items = (int)items;
  const size_t smem = sizeof(float) * (3 * 64 + 9 * 16 + 2 * 16 + 4 * 64 + 16);
  hipLaunchKernelGGL((irb_kernel<3, 1, 3, NT, 1, 1, ACT, true, PF, EAT_ACT_HSWISH, true>),
                     dim3((unsigned)((items + kWaves - 1) / kWaves)), dim3(64 * kWaves), smem, s, a);
  return eat::check_launch("irb_kernel (front)");
}

}  // namespace

namespace {

// Shape -> instantiation of the register-resident block kernel; returns 1 when there is none (the caller's plan then uses
// the separate 1x1 / depthwise kernels), 0 / < 0 after a launch.  dry = only answer the question.
// Tilings measured on MI355X at B = 256 (blocks 2 / 3 / 4 of mn10, ms): one column tile per wave (half the registers: 3-4
// waves per SIMD instead of 2), the next row's B operands a whole row ahead, SE block marching all 5 chunks together:
// 0.36 / 0.28 / 0.25; two tiles per wave 0.40 / 0.31 / 0.34, two tiles + full-row prefetch 0.40 / 0.32 / 0.36, a
// window-swapping variant slower still - those instantiations were removed in round 4 (round-1 LDS-staged kernel:
// 0.65 / 0.51 / 0.40, removed with csrc/mbconv.hip).
int irb_dispatch(IrbArgs a, int k, int stride, int act, bool dry, hipStream_t s) {
  if (act != EAT_ACT_RELU) return 1;
  if (4LL * a.Cin * a.F * a.T >= (1LL << 31) || 4LL * (a.Cexp > a.Cout ? a.Cexp : a.Cout) * a.Fo * a.To >= (1LL << 31)) return 1;   // 32-bit byte offsets inside a sample
  const bool proj = a.wpp != nullptr;
  if (a.Cexp % 8 != 0 || (proj && a.Cout % 4 != 0)) return 1;
  const int MT = (a.Cexp + 15) / 16, MTO = (a.Cout + 15) / 16;
  constexpr int R = EAT_ACT_RELU;
#define EAT_IRB_GO(...) return dry ? 0 : launch_irb<__VA_ARGS__>(a, s)
  if (proj) {
    if (k == 3 && stride == 2 && a.Cin == 16 && MT == 4 && MTO == 2) EAT_IRB_GO(3, 2, 4, 1, 4, 2, R, true, true);
    if (k == 3 && stride == 1 && a.Cin == 24 && MT == 5 && MTO == 2) EAT_IRB_GO(3, 1, 6, 1, 5, 2, R, true, true);
    // (mn10 block 7 - 40 -> 240 -> 80, 3x3 / stride 2, Hardswish - as <3, 2, 10, 1, 15, 5>: the window of 15 chunks plus the
    //  project accumulators need 256 VGPRs + 64 spilled registers; not instantiated)
    return 1;
  }
  if (k == 5 && stride == 2 && a.Cin == 24) {
    if (MT == 5) EAT_IRB_GO(5, 2, 6, 1, 3, 1, R, false, false);
    EAT_IRB_GO(5, 2, 6, 2, 1, 1, R, false, false);
  }
  if (k == 3 && stride == 2 && a.Cin == 16) EAT_IRB_GO(3, 2, 4, 2, 1, 1, R, false, false);
  if (k == 3 && stride == 1 && a.Cin == 24) EAT_IRB_GO(3, 1, 6, 2, 1, 1, R, false, false);
#undef EAT_IRB_GO
  return 1;
}

int block_fused(const float* x, const float* wp_e, const float* bias_e, const float* w_d, const float* bias_d,
                const float* wp_p, const float* bias_p, const float* res, float* y, float* pool, int B, int Cin, int Cexp,
                int Cout, int F, int T, int Fo, int To, int k, int stride, int act, hipStream_t s, const char* who) {
  if (act != EAT_ACT_RELU && act != EAT_ACT_HSWISH) return eat::fail(EAT_EINVAL, "%s: act must be relu/hswish", who);
  const int p = (k - 1) / 2;
  if (Fo != (F + 2 * p - k) / stride + 1 || To != (T + 2 * p - k) / stride + 1)
    return eat::fail(EAT_EINVAL, "%s: output %dx%d inconsistent with input %dx%d", who, Fo, To, F, T);
  IrbArgs a{};
  a.x = x; a.wpe = wp_e; a.bias_e = bias_e; a.wd = w_d; a.bias_d = bias_d; a.wpp = wp_p; a.bias_p = bias_p; a.res = res;
  a.y = y; a.pool = pool;
  a.B = B; a.Cin = Cin; a.Cexp = Cexp; a.Cout = Cout; a.F = F; a.T = T; a.Fo = Fo; a.To = To;
  const int rc = irb_dispatch(a, k, stride, act, false, s);
  if (rc == 1)
    return eat::fail(EAT_EINVAL, "%s: no fused instantiation for Cin=%d Cexp=%d Cout=%d k=%d stride=%d act=%d (ask "
                     "eat_block_fused_supported; use eat_pw_conv_fwd + eat_dw_conv_fwd)", who, Cin, Cexp, Cout, k, stride, act);
  return rc;
}

}  // namespace

// Host query: 1 when eat_mbconv_fwd (proj = 1) / eat_fused_expand_dw_fwd (proj = 0) has an instantiation for the block.
extern "C" int eat_block_fused_supported(int Cin, int Cexp, int Cout, int F, int T, int k, int stride, int act, int proj) {
  IrbArgs a{};
  static const float dummy = 0.0f;
  a.wpp = proj ? &dummy : nullptr;
  const int p = (k - 1) / 2;
  a.B = 1; a.Cin = Cin; a.Cexp = Cexp; a.Cout = Cout; a.F = F; a.T = T;
  a.Fo = (F + 2 * p - k) / stride + 1; a.To = (T + 2 * p - k) / stride + 1;
  return irb_dispatch(a, k, stride, act, true, nullptr) == 0 ? 1 : 0;
}

extern "C" int eat_fused_expand_dw_fwd(const float* x, const float* wp_e, const float* bias_e, const float* w_d,
                                       const float* bias_d, float* y, float* pool, int B, int Cin, int Cexp, int F,
                                       int T, int Fo, int To, int k, int stride, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  return block_fused(x, wp_e, bias_e, w_d, bias_d, nullptr, nullptr, nullptr, y, pool, B, Cin, Cexp, 0, F, T, Fo, To, k,
                     stride, act, (hipStream_t)stream, "eat_fused_expand_dw_fwd");
}

extern "C" int eat_mbconv_fwd(const float* x, const float* wp_e, const float* bias_e, const float* w_d,
                              const float* bias_d, const float* wp_p, const float* bias_p, const float* res, float* y,
                              int B, int Cin, int Cexp, int Cout, int F, int T, int Fo, int To, int k, int stride,
                              int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!wp_p || !bias_p) return eat::fail(EAT_EINVAL, "eat_mbconv_fwd: project weights and bias are required");
  return block_fused(x, wp_e, bias_e, w_d, bias_d, wp_p, bias_p, res, y, nullptr, B, Cin, Cexp, Cout, F, T, Fo, To, k,
                     stride, act, (hipStream_t)stream, "eat_mbconv_fwd");
}

// Network front (stem + first block) on the register-resident kernel (FRONT mode: two column tiles per wave, 0.269 ms for
// mn10 at B = 256; one tile 0.275, round-1 LDS-staged kernel 0.41 - removed with csrc/front.hip).
extern "C" int eat_front_fwd(const float* x, const float* w_s, const float* bias_s, const float* w_d,
                             const float* bias_d, const float* wp_p, const float* bias_p, float* y, int B, int C, int F,
                             int T, int Fo, int To, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (C != 16) return eat::fail(EAT_EINVAL, "eat_front_fwd: C=%d unsupported (the fused front is built for 16 channels)", C);
  if (Fo != (F - 1) / 2 + 1 || To != (T - 1) / 2 + 1)
    return eat::fail(EAT_EINVAL, "eat_front_fwd: output %dx%d does not match input %dx%d", Fo, To, F, T);
  if (act != EAT_ACT_RELU && act != EAT_ACT_HSWISH) return eat::fail(EAT_EINVAL, "eat_front_fwd: act must be relu/hswish");
  if (4LL * F * T >= (1LL << 31) || 4LL * C * Fo * To >= (1LL << 31))
    return eat::fail(EAT_EINVAL, "eat_front_fwd: plane too large for 32-bit byte offsets");
  IrbArgs a{};
  a.x = x; a.wpe = w_s; a.bias_e = bias_s; a.wd = w_d; a.bias_d = bias_d; a.wpp = wp_p; a.bias_p = bias_p; a.res = nullptr;
  a.y = y; a.pool = nullptr;
  a.B = B; a.Cin = 1; a.Cexp = C; a.Cout = C; a.F = Fo; a.T = To; a.Fo = Fo; a.To = To; a.Fm = F; a.Tm = T;
  hipStream_t s = (hipStream_t)stream;
  return act == EAT_ACT_RELU ? launch_front<2, EAT_ACT_RELU, true>(a, s) : launch_front<2, EAT_ACT_HSWISH, true>(a, s);
}
