// Depthwise k x k convolution for the SMALL planes of the late MobileNetV3 blocks (4x32, 8x63, 16x125 at 128 mels /
// 1000 frames; models/mn/block_types.py:150-162), register-resident:
//
//   * one wave owns a whole (b,c) plane (or, for <= 32 columns, the planes of the SAME channel of two consecutive samples
//     side by side in its two half-waves, so that the taps stay wave-uniform); a lane owns
//     CPL consecutive columns.  Every input element is loaded from HBM exactly once, by exactly one lane, as part of a
//     coalesced row segment - the row-ring kernel (conv_spatial.hip) loads each element K times (through L1) and walks
//     4-16 DEPENDENT load rounds down the plane; here all F rows of the plane are requested up front (F*CPL registers)
//     and the next plane group of the wave is already loading while the current one is multiplied;
//   * the horizontal neighbours come from the adjacent lanes by DPP wavefront shifts (wave_shr:1 / wave_shl:1, VALU rate,
//     no LDS): K-1 (stride 1) or <= 3 (stride 2) shifts per input row, shared by the K output rows the row feeds;
//   * taps and bias are wave-uniform (one channel per wave) and live in SGPRs;
//   * the epilogue is conv_spatial.hip's: bias (folded BatchNorm), activation, optional residual add (the stride-1 data
//     gradient of training runs through the same kernel with the taps flipped), per-plane sum for the squeeze of
//     SqueezeExcitation - one atomicAdd per plane, the wave holds the whole plane.
//
// `dw_plane_try` returns 1 when the geometry is not one of the instantiated ones; the caller then uses the row-ring kernel.
#include <cstdlib>
#include "eat_common.h"
#include "act_io.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// value held by the previous / next lane of the plane's lane group; 0 beyond the group (= zero padding of the conv)
template <int LPP>
__device__ __forceinline__ float from_prev(float v, bool first) {
  const float r = dpp0<0x138>(v);                       // wave_shr:1  (lane 0 keeps the 0 of `old`)
  return (LPP < 64 && first) ? 0.0f : r;
}
template <int LPP>
__device__ __forceinline__ float from_next(float v, bool last) {
  const float r = dpp0<0x130>(v);                       // wave_shl:1  (lane 63 keeps 0)
  return (LPP < 64 && last) ? 0.0f : r;
}

// Optional transform of the conv INPUT, evaluated once per loaded element: act_in(in_a[c] * x + in_b[c]) - the train-mode
// BatchNorm + activation of the expand conv, so that the activated tensor is never written (mn_train.py,
// EAT_FUSE_EXPAND_BN).  Zero padding applies to the transformed map: elements outside the plane stay 0.
struct InTf {
  const float* a; const float* b; int act;
};
struct TfCoef { float a, b, lo, ca, cb; };
__device__ __forceinline__ TfCoef tf_coef(const InTf& t, int c) {
  TfCoef k;
  k.a = t.a[c]; k.b = t.b[c];
  k.lo = t.act == EAT_ACT_RELU ? 0.0f : -__builtin_huge_valf();
  k.ca = t.act == EAT_ACT_HSWISH ? (1.0f / 6.0f) : 0.0f;
  k.cb = t.act == EAT_ACT_HSWISH ? 0.5f : 1.0f;
  return k;
}
__device__ __forceinline__ float tf_apply(float v, const TfCoef& k, bool valid) {
  const float u = fmaf(k.a, v, k.b);
  const float y = fmaxf(u, k.lo) * __builtin_amdgcn_fmed3f(fmaf(u, k.ca, k.cb), 0.0f, 1.0f);
  return valid ? y : 0.0f;
}

struct PlaneArgs {
  const float* x; const float* w; const float* bias; const float* res; float* y; float* pool;
  int B, C, T, To, G, flip, act;
  InTf tf;
  eat::DwEpi epi;
  int per_plane_w;
  int b16 = 0;           // 1: x and y are bf16 in HBM (act_io.h), 2: y only (x fp32) - the statistics instance only
};

// d act(u) / du, PyTorch conventions (nn.ReLU / nn.Hardswish backward); `act` is wave-uniform
// (selects between constants only: written as `u < -3 ? 0 : (u <= 3 ? fma : 1)` hipcc emits nested exec-masked branches
// that skip the fma, and every branch splits the basic block the surrounding loads could have been batched in)
__device__ __forceinline__ float act_deriv(float u, int act) {
  if (act == EAT_ACT_RELU) return u > 0.0f ? 1.0f : 0.0f;
  if (act == EAT_ACT_HSWISH) {
    const float m_in = (u >= -3.0f && u <= 3.0f) ? 1.0f : 0.0f, m_hi = u > 3.0f ? 1.0f : 0.0f;
    return fmaf(fmaf(u, 1.0f / 3.0f, 0.5f), m_in, m_hi);
  }
  return 1.0f;
}


// K, S: kernel size / stride; CPL: input columns per lane; LPP: lanes per plane (64, or 32 = two planes per wave);
// F: input rows (compile time: the whole plane is a static register array)
// Raw buffer access (see irb.hip): per-lane byte offset + scalar row offset; a lane whose offset is kOOB loads 0 /
// stores nothing (hardware range check) - no divergent branches around the edge lanes.
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes_to_end) {
  const int n = bytes_to_end < 0x7fffffffLL ? (int)bytes_to_end : 0x7fffffff;   // exact end of the tensor: see fetch()
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
// 8-byte forms (dword-aligned addresses: rows of an odd width start on 4-byte boundaries)
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store2(float v0, float v1, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v0, v1}), r, (int)voff, (int)soff, 0);
}

// Storage-typed buffer access (act_io.h): XT = float or bf16 (the wide tensors of the bf16-storage training plan).  All
// offsets are BYTES (Bio<XT>::kB per element).  A bf16 row of an odd width starts on a 2-byte boundary on every other row:
// the two-column access is ONE dword at a 2-byte aligned address (gfx9 under ROCm runs with unaligned access enabled for
// buffer / global memory).  The last lane of an odd-width row owns one column only; its load is moved back by one element
// (`adj`) so that it never reaches past its row - i.e. never past the end of the tensor - and takes the high half.
template <typename T> struct Bio;
template <> struct Bio<float> {
  static constexpr unsigned kB = 4;
  static __device__ __forceinline__ unsigned adj(unsigned voff, bool) { return voff; }
  static __device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return buf_load(r, voff, soff); }
  static __device__ __forceinline__ f32x2 ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, bool) { return buf_load2(r, voff, soff); }
  static __device__ __forceinline__ void st1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { buf_store(v, r, voff, soff); }
  static __device__ __forceinline__ void st2(float v0, float v1, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { buf_store2(v0, v1, r, voff, soff); }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Bio<eat::bf16_t> {
  static constexpr unsigned kB = 2;
  static __device__ __forceinline__ unsigned adj(unsigned voff, bool part) { return (part && voff != kOOB) ? voff - 2u : voff; }
  static __device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return eat::bf_lo((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0));
  }
  // voff = adj(offset of the lane's first column, part): (first, second) column; a `part` lane has no second column
  static __device__ __forceinline__ f32x2 ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, bool part) {
    const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
    return f32x2{part ? eat::bf_hi(w) : eat::bf_lo(w), part ? 0.0f : eat::bf_hi(w)};
  }
  static __device__ __forceinline__ void st1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(eat::pack_bf2(v, 0.0f) & 0xffffu), r, (int)voff, (int)soff, 0);
  }
  static __device__ __forceinline__ void st2(float v0, float v1, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(eat::pack_bf2(v0, v1), r, (int)voff, (int)soff, 0);
  }
  static __device__ __forceinline__ float rnd(float v) { return eat::bf_round(v); }
};

// PF: keep the next plane group of the wave loading while the current one is multiplied (small planes; big planes have
// enough bytes in flight from the waves of the CU alone and need the registers)
// EPI: 0 plain, 1 residual add (res), 2 training data gradient: the output is multiplied by act'(ga[c] * gz + gb[c]) of
// the tensor gz at the output positions and summed per plane (epi.gpart) - the backward of the BatchNorm + activation
// that FOLLOWS in forward order starts inside the kernel that produces its incoming gradient (mn_train.py);
// STATS: per-plane sum / sum of squares of the output (epi.stats), the BatchNorm batch statistics of the conv output
// PPW: taps per (b,c) plane (DyMN's dynamic depthwise conv in train mode, models/dymn/dy_block.py:103-131): wave-uniform
// scalar loads for one plane per wave, per-lane loads (one address per half-wave) for two
// XT / YT: storage types of x / y (Bio; YT = bf16: the statistics instances of the bf16-storage plan, which sum the ROUNDED
// outputs; XT = float with YT = bf16: the first block, whose depthwise conv reads the fp32 stem output)
template <int K, int S, int CPL, int LPP, int F, bool PF, int ACT, int EPI, bool STATS, bool PPW = false, typename XT = float,
          typename YT = XT>
__global__ __launch_bounds__(256) void dw_plane_kernel(const PlaneArgs a, const float* __restrict__ w_,
                                                       const float* __restrict__ bias_) {
  constexpr int P = (K - 1) / 2;
  constexpr int NPW = 64 / LPP;                          // planes per wave
  constexpr int NE = S == 1 ? CPL + 2 * P : K;           // extended row: input columns CPL*l - P ... as seen by lane l
  constexpr int NO = S == 1 ? CPL : 1;                   // output columns per lane
  constexpr int Fo = (F + 2 * P - K) / S + 1;
  constexpr unsigned EB = Bio<XT>::kB, EBY = Bio<YT>::kB;
  static_assert(EPI == 0 || (EB == 4 && EBY == 4), "residual / derivative epilogues: fp32 storage only");
  const XT* const ax = reinterpret_cast<const XT*>(a.x);
  YT* const ay = reinterpret_cast<YT*>(a.y);
  static_assert(S == 1 || CPL == 2, "stride 2: a lane owns input columns 2l, 2l+1 and output column l");
  const int lane = threadIdx.x & 63;
  const int l = lane & (LPP - 1);
  const int half = NPW == 1 ? 0 : lane / LPP;
  const bool first = l == 0, last = l == LPP - 1;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int T = a.T, To = a.To, C = a.C;
  const int g0 = wave * a.G;                             // first plane group of this wave
  // group g = (sample pair bp, channel c): planes (NPW*bp + half)*C + c
  const int n_groups = ((a.B + NPW - 1) / NPW) * C;
  // per-lane byte offsets inside the group's planes (the half-wave's plane is C planes further on)
  // A lane with two columns moves them as one 8-byte access.  In the last lane of an odd-width row only the first
  // column exists: the LOAD still reads 8 bytes (the second dword is the next row's first element - or, at the very end
  // of the tensor, out of the descriptor's range, which the hardware checks per dword and returns as 0) and the lane
  // zeroes it; the STORE of that lane is a separate dword (vout[1]), the 8-byte store (vout[0]) skips it.
  unsigned vin, vout[NO];
  {
    const unsigned bi = EB * (unsigned)(half * C * (F * T) + CPL * l), bo = EBY * (unsigned)(half * C * (Fo * To) + NO * l);
    vin = CPL * l < T ? bi : kOOB;
    vout[0] = NO * l + NO - 1 < To ? bo : kOOB;
    if (NO == 2) vout[NO - 1] = (NO * l < To && NO * l + 1 >= To) ? bo : kOOB;
  }
  const bool in_part = CPL == 2 && CPL * l + 1 >= T;      // second input column of this lane does not exist
  if (CPL == 2) vin = Bio<XT>::adj(vin, in_part);
  const long long x_elems = (long long)a.B * C * (F * T), y_elems = (long long)a.B * C * (Fo * To);

  float raw[PF ? 2 : 1][F][CPL];
  // first plane of group g (wave-uniform), its channel, and whether this lane's half-wave has a sample
  auto plane_of = [&](int g, int& c, bool& mine) {
    const int gg = g < n_groups ? g : 0;
    const int bp = gg / C;
    c = gg - bp * C;
    mine = g < n_groups && NPW * bp + half < a.B;
    return NPW * bp * C + c;
  };
  auto fetch = [&](int g, float (&r)[F][CPL]) {
    int c; bool mine;
    const int p = plane_of(g, c, mine);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(ax + (size_t)p * (F * T), (long long)EB * (x_elems - (long long)p * (F * T)));
    const unsigned v = mine ? vin : kOOB;
#pragma unroll
    for (int i = 0; i < F; ++i) {
      if constexpr (CPL == 1) {
        r[i][0] = Bio<XT>::ld1(rx, v, EB * (unsigned)(i * T));
      } else {
        const f32x2 pv = Bio<XT>::ld2(rx, v, EB * (unsigned)(i * T), in_part);
        r[i][0] = pv[0];
        r[i][1] = in_part ? 0.0f : pv[1];
      }
    }
  };

  auto compute = [&](int g, float (&r)[F][CPL]) {
    int c; bool mine;
    const int p = plane_of(g, c, mine);
    if (a.tf.a) {                                          // wave-uniform
      const TfCoef tk = tf_coef(a.tf, c);
      const bool v0 = mine && vin != kOOB, v1 = v0 && !in_part;
#pragma unroll
      for (int i = 0; i < F; ++i) {
        r[i][0] = tf_apply(r[i][0], tk, v0);
        if constexpr (CPL == 2) r[i][1] = tf_apply(r[i][1], tk, v1);
      }
    }
    // taps and bias of the (wave-uniform) channel: scalar loads, issued while the rows of the plane are still in flight
    float wk[K * K];
    if constexpr (PPW) {
      const float* wsrc = w_ + (size_t)(mine && NPW > 1 ? p + half * C : p) * (K * K);
#pragma unroll
      for (int i = 0; i < K * K; ++i) wk[i] = wsrc[i];
    } else {
#pragma unroll
      for (int i = 0; i < K * K; ++i) wk[i] = w_[c * (K * K) + i];                 // __restrict__ kernel args: s_load
    }
    if (a.flip) {                                                                   // data gradient: correlate with reversed taps
#pragma unroll
      for (int i = 0; i < K * K / 2; ++i) { const float t = wk[i]; wk[i] = wk[K * K - 1 - i]; wk[K * K - 1 - i] = t; }
    }
    const float b = bias_ ? bias_[c] : 0.0f;
    const long long y_left = (long long)EBY * (y_elems - (long long)p * (Fo * To));
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(ay + (size_t)p * (Fo * To), y_left);
    constexpr bool RES = EPI == 1;
    const __amdgpu_buffer_rsrc_t rr_ = make_rsrc((EPI == 1 ? a.res : EPI == 2 ? a.epi.gz : a.y) + (size_t)p * (Fo * To), y_left);
    const float g_a = EPI == 2 ? a.epi.ga[c] : 0.0f, g_b = EPI == 2 ? a.epi.gb[c] : 0.0f;
    unsigned vo[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) vo[j] = mine ? vout[j] : kOOB;
    const bool has0 = NO * l < To && mine, has1 = NO * l + 1 < To && mine;     // which of the lane's output columns exist
    float ext[F][NE];
    float psum = 0.0f, psq = 0.0f;
#pragma unroll
    for (int i = 0; i < Fo; ++i) {
      __builtin_amdgcn_sched_barrier(0);      // keep the shifted copies of later rows from being hoisted (VGPR pressure)
      // extended rows this output row is the first to need
      const int lo = i == 0 ? 0 : (i - 1) * S + P + 1;
      const int hi = i * S + P < F - 1 ? i * S + P : F - 1;
#pragma unroll
      for (int rr = 0; rr < F; ++rr) {
        if (rr < lo || rr > hi) continue;
#pragma unroll
        for (int t = 0; t < NE; ++t) {
          const int o = t - P;                                      // column offset from CPL*l
          const int q = o >= 0 ? o / CPL : -((-o + CPL - 1) / CPL);  // lane offset
          const int idx = o - q * CPL;
          float v = r[rr][idx];
          if (q == -1) v = from_prev<LPP>(v, first);
          if (q == -2) v = from_prev<LPP>(from_prev<LPP>(v, first), first);
          if (q == 1) v = from_next<LPP>(v, last);
          if (q == 2) v = from_next<LPP>(from_next<LPP>(v, last), last);
          ext[rr][t] = v;
        }
      }
      float acc[NO];
      {
#pragma unroll
        for (int j = 0; j < NO; ++j) acc[j] = b;
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int rr = i * S - P + u;
          if (rr < 0 || rr >= F) continue;
#pragma unroll
          for (int v = 0; v < K; ++v)
#pragma unroll
            for (int j = 0; j < NO; ++j) acc[j] = fmaf(wk[u * K + v], ext[rr][j + v], acc[j]);
        }
      }
      const unsigned so = EBY * (unsigned)(i * To);
      if constexpr (NO == 1) {
        float o = Bio<YT>::rnd(eat::activate<ACT>(acc[0]));
        if constexpr (RES) o += buf_load(rr_, vo[0], so);
        if constexpr (EPI == 2) o *= act_deriv(fmaf(g_a, buf_load(rr_, vo[0], so), g_b), a.epi.gact);
        Bio<YT>::st1(o, ry, vo[0], so);
        psum += has0 ? o : 0.0f;
        if constexpr (STATS) psq += has0 ? o * o : 0.0f;
      } else {
        float o0 = Bio<YT>::rnd(eat::activate<ACT>(acc[0])), o1 = Bio<YT>::rnd(eat::activate<ACT>(acc[1]));
        if constexpr (EPI != 0) {
          const f32x2 rv = buf_load2(rr_, has0 ? 4u * (unsigned)(half * C * (Fo * To) + NO * l) : kOOB, so);
          if constexpr (EPI == 1) {
            o0 += rv[0]; o1 += rv[1];             // (o1 of a lane without a second column is never stored)
          } else {
            o0 *= act_deriv(fmaf(g_a, rv[0], g_b), a.epi.gact);
            o1 *= act_deriv(fmaf(g_a, rv[1], g_b), a.epi.gact);
          }
        }
        Bio<YT>::st2(o0, o1, ry, vo[0], so);
        Bio<YT>::st1(o0, ry, vo[1], so);
        psum += (has0 ? o0 : 0.0f) + (has1 ? o1 : 0.0f);
        if constexpr (STATS) psq += (has0 ? o0 * o0 : 0.0f) + (has1 ? o1 * o1 : 0.0f);
      }
    }
    if (a.pool || STATS || EPI == 2) {
      if (LPP == 64) {
        psum = eat::wave_sum(psum);
        if constexpr (STATS) psq = eat::wave_sum(psq);
      } else {
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) {
          psum += __shfl_xor(psum, o, 64);
          if constexpr (STATS) psq += __shfl_xor(psq, o, 64);
        }
      }
      if (l == 0 && mine) {
        if (a.pool) atomicAdd(a.pool + p + half * C, psum);
        // plain stores: the wave owns the plane (partials [b][2][C] / [b][C], reduced by the finalize kernels)
        const int bsm = (p + half * C) / C;
        if constexpr (STATS) {
          a.epi.stats[((size_t)bsm * 2 + 0) * C + c] = psum;
          a.epi.stats[((size_t)bsm * 2 + 1) * C + c] = psq;
        }
        if constexpr (EPI == 2) a.epi.gpart[(size_t)bsm * C + c] = psum;
      }
    }
  };

  if constexpr (PF) {
    fetch(g0, raw[0]);
    for (int gi = 0; gi < a.G; gi += 2) {
      fetch(g0 + gi + 1, raw[1]);
      compute(g0 + gi, raw[0]);
      if (gi + 2 < a.G) fetch(g0 + gi + 2, raw[0]);
      compute(g0 + gi + 1, raw[1]);
    }
  } else {
    for (int gi = 0; gi < a.G; ++gi) {
      fetch(g0 + gi, raw[0]);
      compute(g0 + gi, raw[0]);
    }
  }
}

template <int K, int S, int CPL, int LPP, int F, bool PF>
int launch_plane(const PlaneArgs& a0, hipStream_t s) {
  PlaneArgs a = a0;
  constexpr int NPW = 64 / LPP;
  const int n_groups = ((a.B + NPW - 1) / NPW) * a.C;
  // plane groups per wave: enough waves to fill the chip several times over, an even count for the 2-deep prefetch
  const int G = PF ? 2 : 1;
  a.G = G;
  const int waves = (n_groups + G - 1) / G;
  const dim3 grid((waves + 3) / 4), blk(256);
  if (a.epi.inner) *a.epi.inner = 1;
  if (a.b16) {                                            // bf16 storage: train-mode conv + statistics (eat_dw_conv_fwd_stats_b16)
    if (!a.epi.stats || a.epi.gz || a.res || a.pool || a.act != EAT_ACT_NONE || a.flip) return 1;
    if (a.per_plane_w) {                                  // ... with per-plane taps (DyMN, eat_dw_conv_dyn_fwd_stats_b16): bf16 -> bf16
      if (a.b16 == 2) return 1;
      hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 0, true, true, eat::bf16_t>), grid, blk, 0, s, a, a.w, a.bias);
      return eat::check_launch("eat_dw_conv_dyn_fwd_stats_b16(plane)");
    }
    if (a.b16 == 2)
      hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 0, true, false, float, eat::bf16_t>), grid, blk, 0, s, a, a.w, a.bias);
    else
      hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 0, true, false, eat::bf16_t>), grid, blk, 0, s, a, a.w, a.bias);
    return eat::check_launch("eat_dw_conv_fwd_stats_b16(plane)");
  }
  if (a.per_plane_w) {                                    // DyMN train mode: plain conv (forward / flipped data gradient) or + statistics
    if (a.epi.gz || a.res || a.act != EAT_ACT_NONE) return 1;
    if (a.epi.stats)
      hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 0, true, true>), grid, blk, 0, s, a, a.w, a.bias);
    else
      hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 0, false, true>), grid, blk, 0, s, a, a.w, a.bias);
    return eat::check_launch("eat_dw_conv_fwd(plane, per-plane taps)");
  }
  if (a.epi.gz) {
    hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 2, false>), grid, blk, 0, s, a, a.w, a.bias);
  } else if (a.epi.stats) {
    hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 0, true>), grid, blk, 0, s, a, a.w, a.bias);
  } else if (a.res) {
    hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, EAT_ACT_NONE, 1, false>), grid, blk, 0, s, a, a.w, a.bias);
  } else {
    EAT_DISPATCH_ACT(a.act, hipLaunchKernelGGL((dw_plane_kernel<K, S, CPL, LPP, F, PF, ACT, 0, false>), grid, blk, 0, s, a, a.w, a.bias));
  }
  return eat::check_launch("eat_dw_conv_fwd(plane)");
}

// ---- the same recipe for LARGE planes: a wave owns a tile of RO output rows x one column strip, with halo lanes / halo
// rows re-read from the neighbouring tiles (L2 hits).  Lane l holds input columns c0 + 2l, c0 + 2l + 1 of every tile row
// (one 8-byte load each, all rows of the tile requested up front); lane 0 (and lane 63 where the filter needs it) are
// halo lanes that produce no output.  Strip width WO (outputs per strip) is balanced by the host: 125 columns for a 3x3 /
// stride-1 conv on 500- or 250-wide planes (4 / 2 strips, no waste), 63 for 3x3 / stride 2, <= 124 / 62 for 5x5.
// per_plane_w: taps per (b,c) plane (DyMN train-mode depthwise conv, models/dymn/dy_block.py:103-131).
struct TileArgs {
  const float* x; const float* res; float* y; float* pool;
  int B, C, F, T, Fo, To, n_rc, n_cs, WO, flip, per_plane_w;
  InTf tf;
  eat::DwEpi epi;
  int b16 = 0;           // x and y are bf16 in HBM (act_io.h): the statistics instance only
};

template <int K, int S, int RO, int ACT, int EPI, bool STATS, typename XT = float, typename YT = XT>
__global__ __launch_bounds__(256) void dw_tile_kernel(const TileArgs a, const float* __restrict__ w_,
                                                      const float* __restrict__ bias_) {
  constexpr int P = (K - 1) / 2, CPL = 2, LPP = 64;
  constexpr int NE = S == 1 ? CPL + 2 * P : K;
  constexpr int NO = S == 1 ? 2 : 1;
  constexpr int FI = (RO - 1) * S + K;                   // input rows under a tile
  constexpr unsigned EB = Bio<XT>::kB, EBY = Bio<YT>::kB;
  static_assert(EPI == 0 || (EB == 4 && EBY == 4), "residual / derivative epilogues: fp32 storage only");
  const int l = threadIdx.x & 63;
  const bool first = l == 0, last = l == 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int tpp = a.n_rc * a.n_cs;
  if (wave >= a.B * a.C * tpp) return;                   // wave-uniform (host: B*C*tpp < 2^31)
  const int p = wave / tpp, t = wave - p * tpp;
  const int rc = t / a.n_cs, cs = t - rc * a.n_cs;
  const int c = p % a.C;
  const int F = a.F, T = a.T, Fo = a.Fo, To = a.To;
  const int o_lo = cs * a.WO, o_hi = (o_lo + a.WO) < To ? (o_lo + a.WO) : To;   // output columns of this strip
  const int c0 = (S == 1 ? o_lo : 2 * o_lo) - 2;         // input column of lane 0 (a halo lane)
  const int col_in = c0 + 2 * l;
  const bool in_part = col_in + 1 >= T;
  const unsigned vin = Bio<XT>::adj((col_in >= 0 && col_in < T) ? EB * (unsigned)col_in : kOOB, in_part);
  // output columns of this lane
  const int oc = S == 1 ? col_in : o_lo - 1 + l;
  const bool ok0 = oc >= o_lo && oc < o_hi, ok1 = NO == 2 && oc + 1 >= o_lo && oc + 1 < o_hi;
  unsigned vout[2];
  vout[0] = (NO == 2 ? (ok0 && ok1) : ok0) ? EBY * (unsigned)oc : kOOB;
  vout[1] = (NO == 2 && ok0 && !ok1) ? EBY * (unsigned)oc : kOOB;
  const int r0o = rc * RO, r0i = r0o * S - P;
  const long long x_left = (long long)EB * ((long long)a.B * a.C - p) * ((long long)F * T);
  const long long y_left = (long long)EBY * ((long long)a.B * a.C - p) * ((long long)Fo * To);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const XT*>(a.x) + (size_t)p * F * T, x_left);
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(reinterpret_cast<YT*>(a.y) + (size_t)p * Fo * To, y_left);
  constexpr bool RES = EPI == 1;
  const __amdgpu_buffer_rsrc_t rr_ = make_rsrc((EPI == 1 ? a.res : EPI == 2 ? a.epi.gz : a.y) + (size_t)p * Fo * To, y_left);
  const float g_a = EPI == 2 ? a.epi.ga[c] : 0.0f, g_b = EPI == 2 ? a.epi.gb[c] : 0.0f;

  float r[FI][CPL];
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int rin = r0i + i;
    const bool rok = rin >= 0 && rin < F;                // wave-uniform: rows above / below the plane read as zero
    const f32x2 pv = Bio<XT>::ld2(rx, rok ? vin : kOOB, rok ? EB * (unsigned)(rin * T) : 0u, in_part);
    r[i][0] = pv[0];
    r[i][1] = in_part ? 0.0f : pv[1];
  }
  if (a.tf.a) {                                            // wave-uniform
    const TfCoef tk = tf_coef(a.tf, c);
    const bool v0 = vin != kOOB, v1 = v0 && !in_part;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const bool rok = r0i + i >= 0 && r0i + i < F;
      r[i][0] = tf_apply(r[i][0], tk, rok && v0);
      r[i][1] = tf_apply(r[i][1], tk, rok && v1);
    }
  }
  float wk[K * K];
  const int wb = a.per_plane_w ? p : c;
#pragma unroll
  for (int i = 0; i < K * K; ++i) wk[i] = w_[(size_t)wb * (K * K) + i];
  if (a.flip) {
#pragma unroll
    for (int i = 0; i < K * K / 2; ++i) { const float tt = wk[i]; wk[i] = wk[K * K - 1 - i]; wk[K * K - 1 - i] = tt; }
  }
  const float b = bias_ ? bias_[c] : 0.0f;

  float ext[FI][NE];
  float psum = 0.0f, psq = 0.0f;
#pragma unroll
  for (int i = 0; i < RO; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    const int lo = i == 0 ? 0 : (i - 1) * S + K;
    const int hi = i * S + K - 1;
#pragma unroll
    for (int rr = 0; rr < FI; ++rr) {
      if (rr < lo || rr > hi) continue;
#pragma unroll
      for (int tt = 0; tt < NE; ++tt) {
        const int o = tt - P;
        const int q = o >= 0 ? o / CPL : -((-o + CPL - 1) / CPL);
        const int idx = o - q * CPL;
        float v = r[rr][idx];
        if (q == -1) v = from_prev<LPP>(v, first);
        if (q == 1) v = from_next<LPP>(v, last);
        ext[rr][tt] = v;
      }
    }
    float acc[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) acc[j] = b;
#pragma unroll
    for (int u = 0; u < K; ++u)
#pragma unroll
      for (int v = 0; v < K; ++v)
#pragma unroll
        for (int j = 0; j < NO; ++j) acc[j] = fmaf(wk[u * K + v], ext[i * S + u][j + v], acc[j]);
    const int ro = r0o + i;
    const bool rowok = ro < Fo;                          // wave-uniform
    const unsigned so = rowok ? EBY * (unsigned)(ro * To) : 0u;
    const unsigned v0 = rowok ? vout[0] : kOOB, v1 = rowok ? vout[1] : kOOB;
    if constexpr (NO == 1) {
      float o = Bio<YT>::rnd(eat::activate<ACT>(acc[0]));
      if constexpr (RES) o += buf_load(rr_, v0, so);
      if constexpr (EPI == 2) o *= act_deriv(fmaf(g_a, buf_load(rr_, v0, so), g_b), a.epi.gact);
      Bio<YT>::st1(o, ry, v0, so);
      psum += (rowok && ok0) ? o : 0.0f;
      if constexpr (STATS) psq += (rowok && ok0) ? o * o : 0.0f;
    } else {
      float o0 = Bio<YT>::rnd(eat::activate<ACT>(acc[0])), o1 = Bio<YT>::rnd(eat::activate<ACT>(acc[1]));
      if constexpr (EPI != 0) {
        const f32x2 rv = buf_load2(rr_, (rowok && ok0) ? 4u * (unsigned)oc : kOOB, so);
        if constexpr (EPI == 1) {
          o0 += rv[0]; o1 += rv[1];
        } else {
          o0 *= act_deriv(fmaf(g_a, rv[0], g_b), a.epi.gact);
          o1 *= act_deriv(fmaf(g_a, rv[1], g_b), a.epi.gact);
        }
      }
      Bio<YT>::st2(o0, o1, ry, v0, so);
      Bio<YT>::st1(o0, ry, v1, so);
      psum += ((rowok && ok0) ? o0 : 0.0f) + ((rowok && ok1) ? o1 : 0.0f);
      if constexpr (STATS) psq += ((rowok && ok0) ? o0 * o0 : 0.0f) + ((rowok && ok1) ? o1 * o1 : 0.0f);
    }
  }
  if (a.pool) {
    const float ps = eat::wave_sum(psum);
    if (l == 0) atomicAdd(a.pool + p, ps);
  }
  // training epilogues: one partial per wave (= per tile), plain stores; layouts [b][2][C][tpp] and [b][C][tpp]
  if constexpr (STATS) {
    psum = eat::wave_sum(psum);
    psq = eat::wave_sum(psq);
    if (l == 0) {
      const int bsm = p / a.C;
      a.epi.stats[(((size_t)bsm * 2 + 0) * a.C + c) * tpp + t] = psum;
      a.epi.stats[(((size_t)bsm * 2 + 1) * a.C + c) * tpp + t] = psq;
    }
  }
  if constexpr (EPI == 2) {
    psum = eat::wave_sum(psum);
    if (l == 0) a.epi.gpart[(size_t)p * tpp + t] = psum;
  }
}

template <int K, int S, int RO>
int launch_tile(TileArgs a, const float* w, const float* bias, int act, hipStream_t s) {
  constexpr int WMAX = S == 1 ? (K == 3 ? 125 : 124) : (K == 3 ? 63 : 62);
  a.n_cs = (a.To + WMAX - 1) / WMAX;
  a.WO = (a.To + a.n_cs - 1) / a.n_cs;                   // balanced strips
  a.n_rc = (a.Fo + RO - 1) / RO;
  const long long waves = (long long)a.B * a.C * a.n_rc * a.n_cs;
  if (waves > 0x7fffffffLL) return 1;
  const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
  if (a.epi.inner) *a.epi.inner = a.n_rc * a.n_cs;
  if (a.b16) {                                            // bf16 storage: train-mode conv + statistics
    if (!a.epi.stats || a.epi.gz || a.res || a.pool || act != EAT_ACT_NONE || a.flip) return 1;
    if (a.b16 == 2)
      hipLaunchKernelGGL((dw_tile_kernel<K, S, RO, EAT_ACT_NONE, 0, true, float, eat::bf16_t>), grid, blk, 0, s, a, w, bias);
    else
      hipLaunchKernelGGL((dw_tile_kernel<K, S, RO, EAT_ACT_NONE, 0, true, eat::bf16_t>), grid, blk, 0, s, a, w, bias);
    return eat::check_launch("eat_dw_conv_fwd_stats_b16(tile)");
  }
  if (a.epi.gz) {
    hipLaunchKernelGGL((dw_tile_kernel<K, S, RO, EAT_ACT_NONE, 2, false>), grid, blk, 0, s, a, w, bias);
  } else if (a.epi.stats) {
    hipLaunchKernelGGL((dw_tile_kernel<K, S, RO, EAT_ACT_NONE, 0, true>), grid, blk, 0, s, a, w, bias);
  } else if (a.res) {
    hipLaunchKernelGGL((dw_tile_kernel<K, S, RO, EAT_ACT_NONE, 1, false>), grid, blk, 0, s, a, w, bias);
  } else {
    EAT_DISPATCH_ACT(act, hipLaunchKernelGGL((dw_tile_kernel<K, S, RO, ACT, 0, false>), grid, blk, 0, s, a, w, bias));
  }
  return eat::check_launch("eat_dw_conv_fwd(tile)");
}

// ---- stride-2 data gradient, tile form: dx[i][j] = sum_{u,v} w[u][v] dz[(i+P-u)/2][(j+P-v)/2] over the (u, v) for which
// both quotients are integers.  Roles reversed with respect to the forward: a lane holds ONE dz column q of every tile row
// (neighbours q-1 / q+1 by DPP) and produces the dx columns 2q, 2q+1 of the rows 2r, 2r+1 - each of the four
// (row parity, column parity) classes has its own compile-time tap subset (K*K/4 FMAs per dx element on average), and
// every dx row segment leaves as one 8-byte store per lane.  Lanes 0 and 63 are halo lanes.
struct TileDgArgs {
  const float* dz; const float* res; float* dx;
  int B, C, F, T, Fo, To, n_rc, n_cs, WO, per_plane_w;
  eat::DwEpi epi;
};

template <int K, int RO, int EPI>
__global__ __launch_bounds__(256) void dw_tile_dgrad2_kernel(const TileDgArgs a, const float* __restrict__ w_) {
  constexpr int P = (K - 1) / 2, LPP = 64;
  constexpr int FI = RO + 2;                             // dz rows of a tile incl. one halo row above and below
  const int l = threadIdx.x & 63;
  const bool first = l == 0, last = l == 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int tpp = a.n_rc * a.n_cs;
  if (wave >= a.B * a.C * tpp) return;
  const int p = wave / tpp, t = wave - p * tpp;
  const int rc = t / a.n_cs, cs = t - rc * a.n_cs;
  const int c = p % a.C;
  const int F = a.F, T = a.T, Fo = a.Fo, To = a.To;
  const int q_lo = cs * a.WO, q_hi = (q_lo + a.WO) < To ? (q_lo + a.WO) : To;
  const int q = q_lo - 1 + l;                            // dz column of this lane
  const unsigned vz = (q >= 0 && q < To) ? 4u * (unsigned)q : kOOB;
  const bool mine = q >= q_lo && q < q_hi;               // this lane produces dx columns 2q, 2q+1
  const bool two = 2 * q + 1 < T;
  const unsigned vx2 = (mine && two) ? 8u * (unsigned)q : kOOB, vx1 = (mine && !two) ? 8u * (unsigned)q : kOOB;
  const int r_lo = rc * RO;
  const long long z_left = 4 * ((long long)a.B * a.C - p) * ((long long)Fo * To);
  const long long x_left = 4 * ((long long)a.B * a.C - p) * ((long long)F * T);
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dz + (size_t)p * Fo * To, z_left);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.dx + (size_t)p * F * T, x_left);
  constexpr bool RES = EPI == 1;
  const __amdgpu_buffer_rsrc_t rr_ = make_rsrc((EPI == 1 ? a.res : EPI == 2 ? a.epi.gz : a.dx) + (size_t)p * F * T, x_left);
  const float g_a = EPI == 2 ? a.epi.ga[c] : 0.0f, g_b = EPI == 2 ? a.epi.gb[c] : 0.0f;
  float psum = 0.0f;
  float z[FI][3];                                        // [row][q-1, q, q+1]
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int r = r_lo - 1 + i;
    const bool rok = r >= 0 && r < Fo;
    z[i][1] = buf_load(rz, rok ? vz : kOOB, rok ? 4u * (unsigned)(r * To) : 0u);
  }
  float wk[K * K];
  const int wb = a.per_plane_w ? p : c;
#pragma unroll
  for (int i = 0; i < K * K; ++i) wk[i] = w_[(size_t)wb * (K * K) + i];
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    z[i][0] = from_prev<LPP>(z[i][1], first);
    z[i][2] = from_next<LPP>(z[i][1], last);
  }
#pragma unroll
  for (int i = 0; i < RO; ++i) {                         // dz row r = r_lo + i  ->  dx rows 2r, 2r+1
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
      float o[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < K; ++u) {
          if ((pa + P - u) & 1) continue;
          const int dr = (pa + P - u) / 2;               // even numerator: exact
#pragma unroll
          for (int v = 0; v < K; ++v) {
            if ((pb + P - v) & 1) continue;
            const int dc = (pb + P - v) / 2;
            acc = fmaf(wk[u * K + v], z[i + 1 + dr][1 + dc], acc);
          }
        }
        o[pb] = acc;
      }
      const int row = 2 * (r_lo + i) + pa;
      const bool rowok = row < F;                        // wave-uniform
      const unsigned so = rowok ? 4u * (unsigned)(row * T) : 0u;
      const unsigned v2 = rowok ? vx2 : kOOB, v1 = rowok ? vx1 : kOOB;
      if constexpr (RES) {
        const f32x2 rv = buf_load2(rr_, v2, so);
        o[0] += rv[0] + buf_load(rr_, v1, so);
        o[1] += rv[1];
      }
      if constexpr (EPI == 2) {
        const f32x2 rv = buf_load2(rr_, v2, so);
        const float z0 = rv[0] + buf_load(rr_, v1, so);     // one of the two loads is out of range (returns 0)
        o[0] *= act_deriv(fmaf(g_a, z0, g_b), a.epi.gact);
        o[1] *= act_deriv(fmaf(g_a, rv[1], g_b), a.epi.gact);
        psum += (rowok && mine) ? (o[0] + (two ? o[1] : 0.0f)) : 0.0f;
      }
      buf_store2(o[0], o[1], rx, v2, so);
      buf_store(o[0], rx, v1, so);
    }
  }
  if constexpr (EPI == 2) {
    psum = eat::wave_sum(psum);
    if (l == 0) a.epi.gpart[(size_t)p * tpp + t] = psum;
  }
}

template <int K>
int launch_tile_dgrad2(TileDgArgs a, const float* w, hipStream_t s) {
  constexpr int RO = 8;
  a.n_cs = (a.To + 61) / 62;
  a.WO = (a.To + a.n_cs - 1) / a.n_cs;
  a.n_rc = (a.Fo + RO - 1) / RO;
  const long long waves = (long long)a.B * a.C * a.n_rc * a.n_cs;
  if (waves > 0x7fffffffLL) return 1;
  const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
  if (a.epi.inner) *a.epi.inner = a.n_rc * a.n_cs;
  if (a.epi.gz) hipLaunchKernelGGL((dw_tile_dgrad2_kernel<K, RO, 2>), grid, blk, 0, s, a, w);
  else if (a.res) hipLaunchKernelGGL((dw_tile_dgrad2_kernel<K, RO, 1>), grid, blk, 0, s, a, w);
  else hipLaunchKernelGGL((dw_tile_dgrad2_kernel<K, RO, 0>), grid, blk, 0, s, a, w);
  return eat::check_launch("eat_dw_conv_dgrad(tile)");
}

// ---- weight gradient on the same ownership: dw[c][u][v] = sum_{b,i,j} dz[b,c,i,j] * x[b,c,i*S+u-P,j*S+v-P] -----------
// One wave walks G plane groups of ONE channel (different samples), each lane accumulating its K*K partial products in
// registers (the extended rows are the forward kernel's: DPP neighbours, vertical taps by register reuse); one
// cross-lane reduction per wave, then K*K atomics - or, per_plane (DyMN: taps are per (b,c), models/dymn/dy_block.py:
// 103-131), one reduction per plane and a plain store of its K*K sums.
struct PlaneWgArgs {
  const float* dz; const float* x; float* dw;
  int B, C, T, To, G, per_plane;
  InTf tf;
};

template <int K, int S, int CPL, int LPP, int F>
__global__ __launch_bounds__(256) void dw_plane_wgrad_kernel(const PlaneWgArgs a) {
  constexpr int P = (K - 1) / 2;
  constexpr int NPW = 64 / LPP;
  constexpr int NE = S == 1 ? CPL + 2 * P : K;
  constexpr int NO = S == 1 ? CPL : 1;                   // dz columns per lane
  constexpr int Fo = (F + 2 * P - K) / S + 1;
  constexpr int KK = K * K;
  const int lane = threadIdx.x & 63;
  const int l = lane & (LPP - 1);
  const int half = NPW == 1 ? 0 : lane / LPP;
  const bool first = l == 0, last = l == LPP - 1;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int T = a.T, To = a.To, C = a.C;
  const int spw = NPW * a.G;                             // samples per wave
  const int n_sg = (a.B + spw - 1) / spw;
  if (wave >= C * n_sg) return;                          // wave-uniform
  const int sg = wave / C, c = wave - sg * C;            // consecutive waves: consecutive channels of the same samples
  unsigned vin, vdz;
  {
    const unsigned bi = 4u * (unsigned)(half * C * (F * T) + CPL * l), bo = 4u * (unsigned)(half * C * (Fo * To) + NO * l);
    vin = CPL * l < T ? bi : kOOB;
    vdz = NO * l < To ? bo : kOOB;
  }
  const bool in_part = CPL == 2 && CPL * l + 1 >= T;
  const bool dz_part = NO == 2 && NO * l + 1 >= To;
  const long long x_elems = (long long)a.B * C * (F * T), z_elems = (long long)a.B * C * (Fo * To);

  float acc[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) acc[i] = 0.0f;

  auto flush = [&](int plane_lo) {
    // sums over the lanes of a plane (per_plane) or of the whole wave (both half-waves hold the same channel)
    float mine_v = 0.0f;
#pragma unroll
    for (int i = 0; i < KK; ++i) {
      float v = acc[i];
#pragma unroll
      for (int o = LPP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (NPW == 2 && !a.per_plane) v += __shfl_xor(v, 32, 64);
      mine_v = l == i ? v : mine_v;
      acc[i] = 0.0f;
    }
    if (a.per_plane) {
      const bool ok = plane_lo >= 0 && (plane_lo / C + half) < a.B;     // plane_lo = b_lo * C + c
      if (l < KK && ok) a.dw[(size_t)(plane_lo + half * C) * KK + l] = mine_v;
    } else if (lane < KK) {
      atomicAdd(a.dw + c * KK + lane, mine_v);
    }
  };

  for (int gi = 0; gi < a.G; ++gi) {
    const int b_lo = (sg * a.G + gi) * NPW;
    if (b_lo >= a.B) break;                               // wave-uniform
    const bool mine = b_lo + half < a.B;
    const int p = b_lo * C + c;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)p * (F * T), 4 * (x_elems - (long long)p * (F * T)));
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dz + (size_t)p * (Fo * To), 4 * (z_elems - (long long)p * (Fo * To)));
    const unsigned vx = mine ? vin : kOOB, vz = mine ? vdz : kOOB;
    float r[F][CPL], d[Fo][NO];
#pragma unroll
    for (int i = 0; i < F; ++i) {
      if constexpr (CPL == 1) {
        r[i][0] = buf_load(rx, vx, 4u * (unsigned)(i * T));
      } else {
        const f32x2 pv = buf_load2(rx, vx, 4u * (unsigned)(i * T));
        r[i][0] = pv[0];
        r[i][1] = in_part ? 0.0f : pv[1];
      }
    }
    if (a.tf.a) {                                          // wave-uniform
      const TfCoef tk = tf_coef(a.tf, c);
      const bool v0 = vx != kOOB, v1 = v0 && !in_part;
#pragma unroll
      for (int i = 0; i < F; ++i) {
        r[i][0] = tf_apply(r[i][0], tk, v0);
        if constexpr (CPL == 2) r[i][1] = tf_apply(r[i][1], tk, v1);
      }
    }
#pragma unroll
    for (int i = 0; i < Fo; ++i) {
      if constexpr (NO == 1) {
        d[i][0] = buf_load(rz, vz, 4u * (unsigned)(i * To));
      } else {
        const f32x2 pv = buf_load2(rz, vz, 4u * (unsigned)(i * To));
        d[i][0] = pv[0];
        d[i][1] = dz_part ? 0.0f : pv[1];
      }
    }
    float ext[F][NE];
#pragma unroll
    for (int i = 0; i < Fo; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      const int lo = i == 0 ? 0 : (i - 1) * S + P + 1;
      const int hi = i * S + P < F - 1 ? i * S + P : F - 1;
#pragma unroll
      for (int rr = 0; rr < F; ++rr) {
        if (rr < lo || rr > hi) continue;
#pragma unroll
        for (int t = 0; t < NE; ++t) {
          const int o = t - P;
          const int q = o >= 0 ? o / CPL : -((-o + CPL - 1) / CPL);
          const int idx = o - q * CPL;
          float v = r[rr][idx];
          if (q == -1) v = from_prev<LPP>(v, first);
          if (q == -2) v = from_prev<LPP>(from_prev<LPP>(v, first), first);
          if (q == 1) v = from_next<LPP>(v, last);
          if (q == 2) v = from_next<LPP>(from_next<LPP>(v, last), last);
          ext[rr][t] = v;
        }
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int rr = i * S - P + u;
        if (rr < 0 || rr >= F) continue;
#pragma unroll
        for (int v = 0; v < K; ++v)
#pragma unroll
          for (int j = 0; j < NO; ++j) acc[u * K + v] = fmaf(d[i][j], ext[rr][j + v], acc[u * K + v]);
      }
    }
    if (a.per_plane) flush(p);
  }
  if (!a.per_plane) flush(-1);
}

// tile form of the weight gradient for large planes: one wave = (channel, tile, G samples)
struct TileWgArgs {
  const float* dz; const float* x; float* dw;
  int B, C, F, T, Fo, To, n_rc, n_cs, WO, G, per_plane;
  InTf tf;
};

template <int K, int S, int RO>
__global__ __launch_bounds__(256) void dw_tile_wgrad_kernel(const TileWgArgs a) {
  constexpr int P = (K - 1) / 2, CPL = 2, LPP = 64;
  constexpr int NE = S == 1 ? CPL + 2 * P : K;
  constexpr int NO = S == 1 ? 2 : 1;
  constexpr int FI = (RO - 1) * S + K;
  constexpr int KK = K * K;
  const int l = threadIdx.x & 63;
  const bool first = l == 0, last = l == 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int tpp = a.n_rc * a.n_cs;
  const int n_sg = (a.B + a.G - 1) / a.G;
  if (wave >= n_sg * a.C * tpp) return;
  // consecutive waves: the tiles of one plane, then the next channel, then the next sample group
  const int pc = wave / tpp, t = wave - pc * tpp;
  const int sg = pc / a.C, c = pc - sg * a.C;
  const int rc = t / a.n_cs, cs = t - rc * a.n_cs;
  const int F = a.F, T = a.T, Fo = a.Fo, To = a.To;
  const int o_lo = cs * a.WO, o_hi = (o_lo + a.WO) < To ? (o_lo + a.WO) : To;
  const int c0 = (S == 1 ? o_lo : 2 * o_lo) - 2;
  const int col_in = c0 + 2 * l;
  const unsigned vin = (col_in >= 0 && col_in < T) ? 4u * (unsigned)col_in : kOOB;
  const bool in_part = col_in + 1 >= T;
  const int oc = S == 1 ? col_in : o_lo - 1 + l;
  const bool ok0 = oc >= o_lo && oc < o_hi, ok1 = NO == 2 && oc + 1 >= o_lo && oc + 1 < o_hi;
  // dz: an 8-byte load needs its first column inside the tensor; a lane whose first column is a halo column but whose
  // second is valid does not occur (halo lanes are whole lanes on the left, and the right edge only drops column 1)
  const unsigned vdz = ok0 ? 4u * (unsigned)oc : kOOB;
  const int r0o = rc * RO, r0i = r0o * S - P;
  float acc[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) acc[i] = 0.0f;

  auto flush = [&](int plane) {
    float mine_v = 0.0f;
#pragma unroll
    for (int i = 0; i < KK; ++i) {
      const float v = eat::wave_sum(acc[i]);
      mine_v = l == i ? v : mine_v;
      acc[i] = 0.0f;
    }
    if (l < KK) atomicAdd(a.dw + (size_t)(a.per_plane ? plane : c) * KK + l, mine_v);
  };

  for (int gi = 0; gi < a.G; ++gi) {
    const int b = sg * a.G + gi;
    if (b >= a.B) break;
    const int p = b * a.C + c;
    const long long x_left = 4 * ((long long)a.B * a.C - p) * ((long long)F * T);
    const long long z_left = 4 * ((long long)a.B * a.C - p) * ((long long)Fo * To);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)p * F * T, x_left);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dz + (size_t)p * Fo * To, z_left);
    float r[FI][CPL], d[RO][NO];
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int rin = r0i + i;
      const bool rok = rin >= 0 && rin < F;
      const f32x2 pv = buf_load2(rx, rok ? vin : kOOB, rok ? 4u * (unsigned)(rin * T) : 0u);
      r[i][0] = pv[0];
      r[i][1] = in_part ? 0.0f : pv[1];
    }
    if (a.tf.a) {                                          // wave-uniform
      const TfCoef tk = tf_coef(a.tf, c);
      const bool v0 = vin != kOOB, v1 = v0 && !in_part;
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const bool rok = r0i + i >= 0 && r0i + i < F;
        r[i][0] = tf_apply(r[i][0], tk, rok && v0);
        r[i][1] = tf_apply(r[i][1], tk, rok && v1);
      }
    }
#pragma unroll
    for (int i = 0; i < RO; ++i) {
      const int ro = r0o + i;
      const bool rok = ro < Fo;
      if constexpr (NO == 1) {
        d[i][0] = buf_load(rz, rok ? vdz : kOOB, rok ? 4u * (unsigned)(ro * To) : 0u);
      } else {
        const f32x2 pv = buf_load2(rz, rok ? vdz : kOOB, rok ? 4u * (unsigned)(ro * To) : 0u);
        d[i][0] = pv[0];
        d[i][1] = ok1 ? pv[1] : 0.0f;
      }
    }
    float ext[FI][NE];
#pragma unroll
    for (int i = 0; i < RO; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      const int lo = i == 0 ? 0 : (i - 1) * S + K;
      const int hi = i * S + K - 1;
#pragma unroll
      for (int rr = 0; rr < FI; ++rr) {
        if (rr < lo || rr > hi) continue;
#pragma unroll
        for (int tt = 0; tt < NE; ++tt) {
          const int o = tt - P;
          const int q = o >= 0 ? o / CPL : -((-o + CPL - 1) / CPL);
          const int idx = o - q * CPL;
          float v = r[rr][idx];
          if (q == -1) v = from_prev<LPP>(v, first);
          if (q == 1) v = from_next<LPP>(v, last);
          ext[rr][tt] = v;
        }
      }
#pragma unroll
      for (int u = 0; u < K; ++u)
#pragma unroll
        for (int v = 0; v < K; ++v)
#pragma unroll
          for (int j = 0; j < NO; ++j) acc[u * K + v] = fmaf(d[i][j], ext[i * S + u][j + v], acc[u * K + v]);
    }
    if (a.per_plane) flush(p);
  }
  if (!a.per_plane) flush(0);
}

template <int K, int S, int RO>
int launch_tile_wgrad(TileWgArgs a, hipStream_t s) {
  constexpr int WMAX = S == 1 ? (K == 3 ? 125 : 124) : (K == 3 ? 63 : 62);
  a.n_cs = (a.To + WMAX - 1) / WMAX;
  a.WO = (a.To + a.n_cs - 1) / a.n_cs;
  a.n_rc = (a.Fo + RO - 1) / RO;
  int G = 1;
  if (!a.per_plane) {
    G = 8;
    while (G > 1 && (long long)a.C * a.n_rc * a.n_cs * ((a.B + G - 1) / G) < 8192) G >>= 1;
  }
  a.G = G;
  const long long waves = (long long)((a.B + G - 1) / G) * a.C * a.n_rc * a.n_cs;
  if (waves > 0x7fffffffLL) return 1;
  hipLaunchKernelGGL((dw_tile_wgrad_kernel<K, S, RO>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  return eat::check_launch("eat_dw_conv_wgrad(tile)");
}

// ---- merged depthwise backward (round 3): weight gradient AND data gradient (+ the activation-derivative epilogue of
// eat_dw_conv_dgrad_g) of one tile from ONE load of dz and of the pre-BN expand output x.  The two separate kernels
// (dw_*_wgrad_kernel, dw_tile_kernel<flip> / dw_tile_dgrad2_kernel) share their ownership - lane l holds x / dx columns
// (c, c+1) and, for stride 2, dz column c/2 - and both read dz and x: 1 pass over the expanded tensor and 1 over the
// depthwise output less per block.  Any plane size (a small plane is one tile); G samples per wave amortise the K*K
// cross-lane reduction of the weight gradient.
//   dw[c][u][v] += sum dz[i][j] * act(a x + b)[i*S+u-P][j*S+v-P]                      (zero padding of the ACTIVATED map)
//   g[i'][j']   = (sum_{u,v} w[u][v] dz[(i'+P-u)/S][(j'+P-v)/S]) * act'(a x[i'][j'] + b),   gpart = sum g per tile
// BN = true: `dz` is the gradient w.r.t. the ACTIVATED BatchNorm output of this conv (times gscale[b,c] plus gadd[b,c]: the
// squeeze-excitation gate) and the BatchNorm + activation backward  dz = a (g - m1 - xhat m2),  g = (d gs + ga) act'(a z + b)
// is evaluated on load from (d, z) with the channel sums of the reduce pass - bn_act_bwd_apply's pass (2 reads + 1 write of
// the conv-output-sized tensor) becomes one more read here.
struct DzBn {
  const float* z; const float* a; const float* b; const float* mean; const float* invstd;
  const float* gscale; const float* gadd; const double* sums; double n; int act; int frozen;
};
struct DwBwdArgs {
  const float* dz; const float* x; float* g; float* dw; float* gpart;
  int B, C, F, T, Fo, To, n_rc, n_cs, WO, G;
  InTf tf;
  DzBn bn;
  const float* res;      // PPW: added to g (the gradient of a skip connection that ends at the conv input), or NULL
  float* gzpart;         // PPW: per-tile partials of sum g * x (x = the raw conv input), layout of gpart, or NULL
  int b16 = 0;           // 1: dz, bn.z, x and g are bf16 in HBM (act_io.h), 2: dz and bn.z only - the BatchNorm-on-load instances
};

// LPP = 64: a wave owns one tile of one plane (column strips with halo lanes).  LPP = 32 / 16 (small planes, T <= 2 LPP): a
// lane group owns the whole row of ITS plane - 64 / LPP samples of the same channel per wave, every lane busy, no halo
// lanes (the zero padding of the conv is the group edge of from_prev / from_next).
// WR with LPP = 64: one plane per wave, rows of <= 128 columns without halo lanes (a 125-column row of a 5 x 5 conv needs two
// strips of the strip mode - 124 columns + 2 halo lanes - with half of the lanes idle in each).
// Sum N values per lane over the lanes of a group: at the step with lane distance O a lane keeps one half of its values and
// hands the other half to lane ^ O (all indices compile-time: a run-time half size turns v[] into a waterfall of
// indexed-register moves); once one value is left the remaining steps are plain sums.  vidx: index of the first value the
// lane ends up with.
template <int NV, int N, int O>
__device__ __forceinline__ void tap_reduce(float (&v)[NV], int l, int& vidx) {
  if constexpr (O > 0) {
    if constexpr (N > 1) {
      constexpr int H = N / 2;
      const bool up = (l & O) != 0;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const float keep = up ? v[i + H] : v[i];
        const float send = up ? v[i] : v[i + H];
        v[i] = keep + __shfl_xor(send, O, 64);
      }
      vidx += up ? H : 0;
      tap_reduce<NV, H, O / 2>(v, l, vidx);
    } else {
      v[0] += __shfl_xor(v[0], O, 64);
      tap_reduce<NV, 1, O / 2>(v, l, vidx);
    }
  }
}

// PPW: taps and weight gradient per (b,c) plane (DyMN's dynamic depthwise conv, models/dymn/dy_block.py:103-131): the taps
// of the lane group's own plane are loaded per sample, the K*K sums are reduced over the lane group after every sample and
// stored (one tile per plane) or added (several) to dw (B, C, K*K)
// XT: storage type of x and g, ZT: of dz and bn.z (Bio; bf16 = the bf16-storage plan: every wide tensor of the block; the
// first block - no expand conv - has an fp32 conv input and hands an fp32 gradient to the stem: XT = float, ZT = bf16)
template <int K, int S, int RO, bool BN, int LPP, bool WR, bool PPW = false, typename XT = float, typename ZT = XT>
__global__ __launch_bounds__(256) void dw_bwd_tile_kernel(const DwBwdArgs a, const float* __restrict__ w_) {
  constexpr int P = (K - 1) / 2, KK = K * K, NPW = 64 / LPP;
  static_assert(WR || LPP == 64, "strip mode owns the whole wave");
  constexpr int FX = S == 1 ? RO + 2 * P : 2 * RO + 2 * P - 1;     // x rows of a tile (with halo)
  constexpr int FD = S == 1 ? RO + 2 * P : RO + 2;                 // dz rows of a tile (with halo)
  constexpr int DOFF = S == 1 ? P : 1;                             // dz-array index of the tile's first dz row
  constexpr int ND = S == 1 ? 2 : 1;                               // dz columns per lane
  constexpr int NE = S == 1 ? 2 + 2 * P : K;                       // extended x row: columns under the filter
  constexpr unsigned EB = Bio<XT>::kB, EBZ = Bio<ZT>::kB;
  // (PPW with bf16 storage: the DyMN blocks of the bf16-storage plan; a.res - fp32 - only with XT = float, host-checked)
  const int lane = threadIdx.x & 63;
  const int l = lane & (LPP - 1);
  const int half = lane / LPP;                                     // which of the wave's NPW planes (samples)
  const bool first = l == 0, last = l == LPP - 1;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int tpp = a.n_rc * a.n_cs;
  const int n_sg = ((a.B + NPW - 1) / NPW + a.G - 1) / a.G;        // groups of G x NPW samples
  if (wave >= n_sg * a.C * tpp) return;
  const int pc = wave / tpp, t = wave - pc * tpp;
  const int sg = pc / a.C, c = pc - sg * a.C;
  const int rc = t / a.n_cs, cs = t - rc * a.n_cs;
  const int F = a.F, T = a.T, Fo = a.Fo, To = a.To;
  // strip: S == 1 over dx (= x) columns, S == 2 over dz columns; lane 0 (and 63) are halo lanes
  const int s_lo = WR ? 0 : cs * a.WO;
  const int s_hi = WR ? (S == 1 ? T : To) : ((s_lo + a.WO) < (S == 1 ? T : To) ? (s_lo + a.WO) : (S == 1 ? T : To));
  const int q = WR ? l : s_lo - 1 + l;                              // S == 2: dz column of this lane
  const int col_in = S == 1 ? (WR ? 2 * l : s_lo - 2 + 2 * l) : 2 * q;   // first of the lane's two x / dx columns
  // byte offsets inside the wave's first plane; the lane group's own plane lies `half` samples (C planes each) further on
  const unsigned hx = EB * (unsigned)(half * a.C * (F * T)), hz = EBZ * (unsigned)(half * a.C * (Fo * To));
  const bool in_part = col_in + 1 >= T;
  // (bf16: the one-column lane at the end of an odd-width row loads the dword that ENDS with its column - Bio::adj)
  const unsigned vin_b = Bio<XT>::adj((col_in >= 0 && col_in < T) ? hx + EB * (unsigned)col_in : kOOB, in_part);
  // which of the lane's positions belong to the strip (produce output / contribute to the weight gradient)
  const bool ok0_b = S == 1 ? (col_in >= s_lo && col_in < s_hi) : (q >= s_lo && q < s_hi);
  const bool ok1_b = S == 1 ? (col_in + 1 >= s_lo && col_in + 1 < s_hi) : (ok0_b && 2 * q + 1 < T);
  // (stride 1: dz has the geometry of x - the same column pair, in dz's own element size)
  const unsigned vdz_b = S == 1 ? Bio<ZT>::adj((col_in >= 0 && col_in < T) ? hz + EBZ * (unsigned)col_in : kOOB, in_part)
                                : ((q >= 0 && q < To) ? hz + EBZ * (unsigned)q : kOOB);
  const int r0 = rc * RO;                                           // first dx row (S == 1) / dz row (S == 2) of the tile
  const int x0 = S == 1 ? r0 - P : 2 * r0 - P;                      // global row of x-array index 0
  const int d0 = r0 - DOFF;                                         // global row of dz-array index 0

  float wk[KK];
  if constexpr (!PPW) {
#pragma unroll
    for (int i = 0; i < KK; ++i) wk[i] = w_[(size_t)c * KK + i];
  }
  const TfCoef tk = tf_coef(a.tf, c);
  float acc[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) acc[i] = 0.0f;
  // BatchNorm-backward constants of channel c:  dz = za g - zc2 z + zc3
  float za = 1.0f, zb = 0.0f, zc2 = 0.0f, zc3 = 0.0f;
  if constexpr (BN) {
    za = a.bn.a[c]; zb = a.bn.b[c];
    if (!a.bn.frozen) {
      const float mu = a.bn.mean[c], is = a.bn.invstd[c];
      const float m1 = (float)(a.bn.sums[c] / a.bn.n), m2 = (float)(a.bn.sums[a.C + c] / a.bn.n);
      zc2 = za * is * m2;
      zc3 = za * (mu * is * m2 - m1);
    }
  }

  for (int gi = 0; gi < a.G; ++gi) {
    const int b0 = (sg * a.G + gi) * NPW;                           // first sample of the wave's NPW
    if (b0 >= a.B) break;                                           // wave-uniform
    const bool mine = b0 + half < a.B;                              // this lane group has a sample
    const int p = b0 * a.C + c;                                     // plane of lane group 0 (descriptor base)
    const unsigned vin = mine ? vin_b : kOOB, vdz = mine ? vdz_b : kOOB;
    const bool v0 = vin != kOOB, v1 = v0 && !in_part;
    const bool ok0 = ok0_b && mine, ok1 = ok1_b && mine;
    // dx stores: an 8-byte store when both columns exist, else a single dword for the first
    const unsigned vo2 = (ok0 && ok1) ? hx + EB * (unsigned)col_in : kOOB, vo1 = (ok0 && !ok1) ? hx + EB * (unsigned)col_in : kOOB;
    const long long x_left = (long long)EB * ((long long)a.B * a.C - p) * ((long long)F * T);
    const long long z_left = (long long)EBZ * ((long long)a.B * a.C - p) * ((long long)Fo * To);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const XT*>(a.x) + (size_t)p * F * T, x_left);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(reinterpret_cast<const ZT*>(a.dz) + (size_t)p * Fo * To, z_left);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(reinterpret_cast<XT*>(a.g) + (size_t)p * F * T, x_left);
    if constexpr (PPW) {
      const float* wsrc = w_ + (size_t)(mine ? p + half * a.C : p) * KK;
#pragma unroll
      for (int i = 0; i < KK; ++i) { wk[i] = wsrc[i]; acc[i] = 0.0f; }
    }
    float xu[FX][2], dd[FD][ND];
#pragma unroll
    for (int i = 0; i < FX; ++i) {
      const int rin = x0 + i;
      const bool rok = rin >= 0 && rin < F;                         // wave-uniform
      const f32x2 pv = Bio<XT>::ld2(rx, rok ? vin : kOOB, rok ? EB * (unsigned)(rin * T) : 0u, in_part);
      xu[i][0] = pv[0];
      xu[i][1] = pv[1];
    }
    // PPW: the raw conv input at the tile's own (output) positions, for sum g * x
    constexpr int NZR = PPW ? (S == 1 ? RO : 2 * RO) : 1;
    float xraw[NZR][2];
    if constexpr (PPW) {
#pragma unroll
      for (int i = 0; i < NZR; ++i) { xraw[i][0] = xu[i + P][0]; xraw[i][1] = xu[i + P][1]; }
    }
    if constexpr (BN) {
      const __amdgpu_buffer_rsrc_t rzz = make_rsrc(reinterpret_cast<const ZT*>(a.bn.z) + (size_t)p * Fo * To, z_left);
      const int pm = mine ? p + half * a.C : p;                     // this lane group's plane (per-plane SE constants)
      const float gs = a.bn.gscale ? a.bn.gscale[pm] : 1.0f, ga = a.bn.gadd ? a.bn.gadd[pm] : 0.0f;
      // validity as a 0 / 1 factor (operands are 0 outside the plane, so every term is finite): a select around dzf would
      // pull the loads into exec-masked blocks with a wait each
      const float m0 = eat::opaque((S == 1 ? v0 : vdz != kOOB) ? 1.0f : 0.0f), m1v = eat::opaque((S == 1 && v1) ? 1.0f : 0.0f);
      const bool hs = a.bn.act == EAT_ACT_HSWISH, re = a.bn.act == EAT_ACT_RELU;
      auto dzf = [&](float d, float v, float m) {
        const float u = fmaf(za, v, zb);
        const float m_in = (u >= -3.0f && u <= 3.0f) ? 1.0f : 0.0f, m_hi = u > 3.0f ? 1.0f : 0.0f;
        const float dhs = fmaf(fmaf(u, 1.0f / 3.0f, 0.5f), m_in, m_hi);
        const float dre = u > 0.0f ? 1.0f : 0.0f;
        const float gg = fmaf(d, gs, ga) * (hs ? dhs : (re ? dre : 1.0f));
        return fmaf(za, gg, fmaf(-zc2, v, zc3)) * m;
      };
#pragma unroll
      for (int i = 0; i < FD; ++i) {
        const int rin = d0 + i;
        const bool rok = rin >= 0 && rin < Fo;                    // wave-uniform
        const unsigned so = rok ? EBZ * (unsigned)(rin * To) : 0u;
        if constexpr (ND == 2) {
          const f32x2 pv = Bio<ZT>::ld2(rz, rok ? vdz : kOOB, so, in_part);
          const f32x2 zv = Bio<ZT>::ld2(rzz, rok ? vdz : kOOB, so, in_part);
          const float mr = eat::opaque(rok ? 1.0f : 0.0f);
          dd[i][0] = dzf(pv[0], zv[0], m0 * mr);
          dd[i][1] = dzf(pv[1], zv[1], m1v * mr);
        } else {
          const float pv = Bio<ZT>::ld1(rz, rok ? vdz : kOOB, so);
          const float zv = Bio<ZT>::ld1(rzz, rok ? vdz : kOOB, so);
          dd[i][0] = dzf(pv, zv, m0 * eat::opaque(rok ? 1.0f : 0.0f));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < FD; ++i) {
        const int rin = d0 + i;
        const bool rok = rin >= 0 && rin < Fo;
        if constexpr (ND == 2) {
          const f32x2 pv = Bio<ZT>::ld2(rz, rok ? vdz : kOOB, rok ? EBZ * (unsigned)(rin * To) : 0u, in_part);
          dd[i][0] = pv[0];
          dd[i][1] = in_part ? 0.0f : pv[1];
        } else {
          dd[i][0] = Bio<ZT>::ld1(rz, rok ? vdz : kOOB, rok ? EBZ * (unsigned)(rin * To) : 0u);
        }
      }
    }
    // u = a x + b inside the plane, 0 outside: act(0) = 0 for every supported activation, i.e. the zero padding of the
    // ACTIVATED map; y = act(u) is evaluated where the extended rows are built, act'(u) in the epilogue
#pragma unroll
    for (int i = 0; i < FX; ++i) {
      const bool rok = x0 + i >= 0 && x0 + i < F;
      xu[i][0] = (rok && v0) ? fmaf(tk.a, xu[i][0], tk.b) : 0.0f;
      xu[i][1] = (rok && v1) ? fmaf(tk.a, xu[i][1], tk.b) : 0.0f;
    }

    // ---- weight gradient: dz of the tile's own rows / strip columns times the extended rows of y
    {
      float ext[FX][NE];
#pragma unroll
      for (int i = 0; i < RO; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const int lo = i == 0 ? 0 : (i - 1) * S + K;
        const int hi = i * S + K - 1;
#pragma unroll
        for (int rr = 0; rr < FX; ++rr) {
          if (rr < lo || rr > hi) continue;
#pragma unroll
          for (int tt = 0; tt < NE; ++tt) {
            const int o = tt - P;
            const int qq = o >= 0 ? o / 2 : -((-o + 1) / 2);
            const int idx = o - qq * 2;
            float v = xu[rr][idx];
            v = fmaxf(v, tk.lo) * __builtin_amdgcn_fmed3f(fmaf(v, tk.ca, tk.cb), 0.0f, 1.0f);      // y = act(u)
            if (qq == -1) v = from_prev<LPP>(v, first);
            if (qq == 1) v = from_next<LPP>(v, last);
            ext[rr][tt] = v;
          }
        }
        const bool rowok = r0 + i < Fo;                             // wave-uniform (S == 1: Fo == F)
        float dm[ND];
        dm[0] = (rowok && ok0) ? dd[i + DOFF][0] : 0.0f;
        if constexpr (ND == 2) dm[1] = (rowok && ok1) ? dd[i + DOFF][1] : 0.0f;
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
          for (int v = 0; v < K; ++v)
#pragma unroll
            for (int j = 0; j < ND; ++j) acc[u * K + v] = fmaf(dm[j], ext[i * S + u][j + v], acc[u * K + v]);
      }
    }

    // ---- data gradient of the tile + derivative epilogue
    float psum = 0.0f, pgz = 0.0f;
    const __amdgpu_buffer_rsrc_t rres = make_rsrc((PPW && a.res ? a.res : a.g) + (size_t)p * F * T, x_left);
    if constexpr (S == 1) {
      float ext[FD][NE];
#pragma unroll
      for (int i = 0; i < RO; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const int lo = i == 0 ? 0 : (i - 1) + K;
        const int hi = i + K - 1;
#pragma unroll
        for (int rr = 0; rr < FD; ++rr) {
          if (rr < lo || rr > hi) continue;
#pragma unroll
          for (int tt = 0; tt < NE; ++tt) {
            const int o = tt - P;
            const int qq = o >= 0 ? o / 2 : -((-o + 1) / 2);
            const int idx = o - qq * 2;
            float v = dd[rr][idx];
            if (qq == -1) v = from_prev<LPP>(v, first);
            if (qq == 1) v = from_next<LPP>(v, last);
            ext[rr][tt] = v;
          }
        }
        float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
          for (int v = 0; v < K; ++v) {                             // correlation with the flipped taps
            o0 = fmaf(wk[KK - 1 - (u * K + v)], ext[i + u][v], o0);
            o1 = fmaf(wk[KK - 1 - (u * K + v)], ext[i + u][1 + v], o1);
          }
        o0 *= act_deriv(xu[i + P][0], a.tf.act);
        o1 *= act_deriv(xu[i + P][1], a.tf.act);
        const int row = r0 + i;
        const bool rowok = row < F;
        const unsigned so = rowok ? EB * (unsigned)(row * T) : 0u;
        o0 = Bio<XT>::rnd(o0); o1 = Bio<XT>::rnd(o1);                 // (bf16: sum g of the values as stored)
        if constexpr (PPW) {
          psum += ((rowok && ok0) ? o0 : 0.0f) + ((rowok && ok1) ? o1 : 0.0f);
          pgz += ((rowok && ok0) ? o0 * xraw[i][0] : 0.0f) + ((rowok && ok1) ? o1 * xraw[i][1] : 0.0f);
          if (a.res) {                                            // wave-uniform
            const f32x2 rv = buf_load2(rres, (rowok && ok0) ? hx + 4u * (unsigned)col_in : kOOB, so);
            o0 += rv[0]; o1 += rv[1];
          }
          Bio<XT>::st2(o0, o1, rg, rowok ? vo2 : kOOB, so);
          Bio<XT>::st1(o0, rg, rowok ? vo1 : kOOB, so);
        } else {
          Bio<XT>::st2(o0, o1, rg, rowok ? vo2 : kOOB, so);
          Bio<XT>::st1(o0, rg, rowok ? vo1 : kOOB, so);
          psum += ((rowok && ok0) ? o0 : 0.0f) + ((rowok && ok1) ? o1 : 0.0f);
        }
      }
    } else {
      float z[FD][3];
#pragma unroll
      for (int i = 0; i < FD; ++i) {
        z[i][1] = dd[i][0];
        z[i][0] = from_prev<LPP>(dd[i][0], first);
        z[i][2] = from_next<LPP>(dd[i][0], last);
      }
#pragma unroll
      for (int i = 0; i < RO; ++i) {                                // dz row r0 + i -> dx rows 2 (r0 + i), + 1
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
          float o[2];
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            float sacc = 0.0f;
#pragma unroll
            for (int u = 0; u < K; ++u) {
              if ((pa + P - u) & 1) continue;
              const int dr = (pa + P - u) / 2;
#pragma unroll
              for (int v = 0; v < K; ++v) {
                if ((pb + P - v) & 1) continue;
                const int dc = (pb + P - v) / 2;
                sacc = fmaf(wk[u * K + v], z[i + 1 + dr][1 + dc], sacc);
              }
            }
            o[pb] = Bio<XT>::rnd(sacc * act_deriv(xu[2 * i + pa + P][pb], a.tf.act));
          }
          const int row = 2 * (r0 + i) + pa;
          const bool rowok = row < F;
          const unsigned so = rowok ? EB * (unsigned)(row * T) : 0u;
          psum += ((rowok && ok0) ? o[0] : 0.0f) + ((rowok && ok1) ? o[1] : 0.0f);
          if constexpr (PPW) {
            pgz += ((rowok && ok0) ? o[0] * xraw[2 * i + pa][0] : 0.0f) + ((rowok && ok1) ? o[1] * xraw[2 * i + pa][1] : 0.0f);
            if (a.res) {                                          // wave-uniform
              const f32x2 rv = buf_load2(rres, (rowok && ok0) ? hx + 4u * (unsigned)col_in : kOOB, so);
              o[0] += rv[0]; o[1] += rv[1];
            }
          }
          Bio<XT>::st2(o[0], o[1], rg, rowok ? vo2 : kOOB, so);
          Bio<XT>::st1(o[0], rg, rowok ? vo1 : kOOB, so);
        }
      }
    }
    if (a.gpart) {
      if constexpr (LPP < 64) {
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
      } else {
        psum = eat::wave_sum(psum);
      }
      if (l == 0 && mine) a.gpart[(size_t)(p + half * a.C) * tpp + t] = psum;
    }
    if constexpr (PPW) {
      if (a.gzpart) {
        if constexpr (LPP < 64) {
#pragma unroll
          for (int o = LPP >> 1; o > 0; o >>= 1) pgz += __shfl_xor(pgz, o, 64);
        } else {
          pgz = eat::wave_sum(pgz);
        }
        if (l == 0 && mine) a.gzpart[(size_t)(p + half * a.C) * tpp + t] = pgz;
      }
      // the plane's K*K sums over the lane group by a halving butterfly (tap_reduce): 32 shuffles for 25 sums over 64
      // lanes instead of 25 x 6; afterwards lane l holds NFIN consecutive values starting at index vidx
      constexpr int NV0 = KK <= 16 ? 16 : 32;                       // padded value count (power of two)
      constexpr int NFIN = NV0 > LPP ? NV0 / LPP : 1;               // values left per lane (LPP = 16 with 25 taps: 2)
      constexpr int DUP = NV0 >= LPP ? 0 : LPP / NV0 - 1;           // lanes l, l ^ d (d & DUP) end up with the same value
      float v[NV0];
#pragma unroll
      for (int i = 0; i < NV0; ++i) v[i] = i < KK ? acc[i] : 0.0f;
      int vidx = 0;
      tap_reduce<NV0, NV0, (LPP >> 1)>(v, l, vidx);
      float* d = a.dw + (size_t)(mine ? p + half * a.C : p) * KK;
      const bool writer = mine && (l & DUP) == 0;
#pragma unroll
      for (int i = 0; i < NFIN; ++i) {
        const int vi = vidx + i;
        if (writer && vi < KK) {
          if (tpp == 1) d[vi] = v[i]; else atomicAdd(d + vi, v[i]);
        }
      }
    }
  }
  if constexpr (!PPW) {
    // one cross-lane reduction of the K*K weight-gradient sums per wave, then K*K atomics
    float mine_v = 0.0f;
#pragma unroll
    for (int i = 0; i < KK; ++i) {
      const float v = eat::wave_sum(acc[i]);
      mine_v = lane == i ? v : mine_v;
    }
    if (lane < KK) atomicAdd(a.dw + (size_t)c * KK + lane, mine_v);
  }
}

template <int K, int S, int RO>
int launch_dw_bwd(DwBwdArgs a, const float* w, int* h_inner, hipStream_t s, bool ppw = false) {
  const bool bn = a.bn.z != nullptr;
  constexpr int WMAX = S == 1 ? (K == 3 ? 125 : 124) : 62;
  const int n_cols = S == 1 ? a.T : a.To, n_rows = S == 1 ? a.F : a.Fo;
  // small planes (BN instances only): whole rows per lane group, 2 or 4 samples per wave; up to 128 columns: one plane per
  // wave without halo lanes
  const int lpp = !bn ? 64 : (a.T <= 32 ? 16 : (a.T <= 64 ? 32 : 64));
  const bool wr = lpp < 64 || (bn && a.T <= 128);
  const int npw = 64 / lpp;
  a.n_cs = wr ? 1 : (n_cols + WMAX - 1) / WMAX;
  a.WO = (n_cols + a.n_cs - 1) / a.n_cs;
  a.n_rc = (n_rows + RO - 1) / RO;
  const int nb = (a.B + npw - 1) / npw;
  int G = 8;
  while (G > 1 && (long long)a.C * a.n_rc * a.n_cs * ((nb + G - 1) / G) < 8192) G >>= 1;
  a.G = G;
  const long long waves = (long long)((nb + G - 1) / G) * a.C * a.n_rc * a.n_cs;
  if (waves > 0x7fffffffLL) return 1;
  if (h_inner) *h_inner = a.n_rc * a.n_cs;
  const dim3 grid((unsigned)((waves + 3) / 4));
  if (ppw && a.b16) {                                    // per-plane taps with bf16 storage (eat_dw_conv_dyn_bwd_bn_g_b16)
    using BT = eat::bf16_t;
    if (!bn) return 1;
    if (a.b16 == 2) {                                     // x and g fp32 (the block without expand conv: 3x3 / stride 1 on the stem planes)
      if constexpr (K == 3 && S == 1 && RO == 8) {
        if (lpp == 64 && !wr) {
          hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, false, true, float, BT>), grid, dim3(256), 0, s, a, w);
          return eat::check_launch("eat_dw_conv_dyn_bwd_bn_g_b16");
        }
      }
      return 1;
    }
    if (a.res) return 1;                                  // (the fp32 skip gradient goes with an fp32 g)
    if (lpp == 64 && wr) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, true, true, BT, BT>), grid, dim3(256), 0, s, a, w);
    else if (lpp == 64) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, false, true, BT, BT>), grid, dim3(256), 0, s, a, w);
    else if (lpp == 32) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 32, true, true, BT, BT>), grid, dim3(256), 0, s, a, w);
    else hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 16, true, true, BT, BT>), grid, dim3(256), 0, s, a, w);
    return eat::check_launch("eat_dw_conv_dyn_bwd_bn_g_b16");
  }
  if (ppw) {                                             // per-plane taps (DyMN): the BatchNorm-on-load instances only
    if (!bn) return 1;
    if (lpp == 64 && wr) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, true, true>), grid, dim3(256), 0, s, a, w);
    else if (lpp == 64) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, false, true>), grid, dim3(256), 0, s, a, w);
    else if (lpp == 32) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 32, true, true>), grid, dim3(256), 0, s, a, w);
    else hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 16, true, true>), grid, dim3(256), 0, s, a, w);
    return eat::check_launch("eat_dw_conv_dyn_bwd_bn_g");
  }
  if (a.b16) {
    using BT = eat::bf16_t;
    if (!bn) return 1;
#define EAT_BWD16(XT_)                                                                                                          \
    do {                                                                                                                          \
      if (lpp == 64 && wr) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, true, false, XT_, BT>), grid, dim3(256), 0, s, a, w);  \
      else if (lpp == 64) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, false, false, XT_, BT>), grid, dim3(256), 0, s, a, w); \
      else if (lpp == 32) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 32, true, false, XT_, BT>), grid, dim3(256), 0, s, a, w);  \
      else hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 16, true, false, XT_, BT>), grid, dim3(256), 0, s, a, w);                \
    } while (0)
    if (a.b16 == 2) EAT_BWD16(float); else EAT_BWD16(BT);          // 2: x and g fp32 (a block without expand conv)
#undef EAT_BWD16
    return eat::check_launch("eat_dw_conv_bwd_bn_g_b16");
  }
  if (!bn) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, false, 64, false>), grid, dim3(256), 0, s, a, w);
  else if (lpp == 64 && wr) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, true>), grid, dim3(256), 0, s, a, w);
  else if (lpp == 64) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 64, false>), grid, dim3(256), 0, s, a, w);
  else if (lpp == 32) hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 32, true>), grid, dim3(256), 0, s, a, w);
  else hipLaunchKernelGGL((dw_bwd_tile_kernel<K, S, RO, true, 16, true>), grid, dim3(256), 0, s, a, w);
  return eat::check_launch("eat_dw_conv_bwd_g");
}

template <int K, int S, int CPL, int LPP, int F>
int launch_plane_wgrad(const PlaneWgArgs& a0, hipStream_t s) {
  PlaneWgArgs a = a0;
  constexpr int NPW = 64 / LPP;
  // samples per wave: every wave should multiply several planes before its K*K-value reduction, but keep >= ~8 k waves
  int G = 1;
  if (!a.per_plane) {
    G = 8;
    while (G > 1 && (long long)a.C * ((a.B + NPW * G - 1) / (NPW * G)) < 8192) G >>= 1;
  }
  a.G = G;
  const long long waves = (long long)a.C * ((a.B + NPW * G - 1) / (NPW * G));
  hipLaunchKernelGGL((dw_plane_wgrad_kernel<K, S, CPL, LPP, F>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  return eat::check_launch("eat_dw_conv_wgrad(plane)");
}

}  // namespace

namespace eat {

int dw_plane_try(const float* x, const float* w, const float* bias, const float* res, float* y, float* pool, int B, int C,
                 int F, int T, int Fo, int To, int k, int stride, int act, int flip, int per_plane_w, const float* in_a,
                 const float* in_b, int in_act, hipStream_t s, const DwEpi* epi_, int b16) {
  const long long n_planes = (long long)B * C;
  if (n_planes > 0x3fffffffLL) return 1;                 // plane bases are 64-bit, offsets inside a plane 32-bit
  if (res && act != EAT_ACT_NONE) return 1;              // residual add: the data-gradient form only
  const DwEpi epi = epi_ ? *epi_ : DwEpi{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  if ((epi.stats || epi.gz) && (res || pool || act != EAT_ACT_NONE)) return 1;   // training epilogues: plain conv only
  if (epi.gz && per_plane_w) return 1;
  if (T > 128 && (long long)F * T < (1 << 28)) {            // large planes: tiles of rows x column strips
    TileArgs ta{x, res, y, pool, B, C, F, T, Fo, To, 0, 0, 0, flip, per_plane_w, InTf{in_a, in_b, in_act}, epi, b16};
    if (k == 3 && stride == 1) return launch_tile<3, 1, 16>(ta, w, bias, act, s);
    if (k == 5 && stride == 1) return launch_tile<5, 1, 16>(ta, w, bias, act, s);
    if (k == 3 && stride == 2) return launch_tile<3, 2, 8>(ta, w, bias, act, s);
    if (k == 5 && stride == 2) return launch_tile<5, 2, 8>(ta, w, bias, act, s);
  }
  if (per_plane_w && (res || pool || act != EAT_ACT_NONE)) return 1;     // per-plane taps: the plain conv (+ statistics) only
  PlaneArgs a{x, w, bias, res, y, pool, B, C, T, To, 2, flip, act, InTf{in_a, in_b, in_act}, epi, per_plane_w, b16};
  if (k == 3 && stride == 1 && F == 8 && T > 32 && T <= 64) return launch_plane<3, 1, 1, 64, 8, true>(a, s);
  // (no prefetch of the next plane group here: 2 x 16 rows x 2 columns of registers cost the occupancy it would buy)
  if (k == 5 && stride == 1 && F == 16 && T > 64 && T <= 128) return launch_plane<5, 1, 2, 64, 16, false>(a, s);
  if (k == 5 && stride == 2 && F == 8 && T > 32 && T <= 64) return launch_plane<5, 2, 2, 32, 8, true>(a, s);
  if (k == 3 && stride == 2 && F == 16 && T > 64 && T <= 128) return launch_plane<3, 2, 2, 64, 16, true>(a, s);
  if (k == 5 && stride == 1 && F == 4 && T <= 32) return launch_plane<5, 1, 1, 32, 4, true>(a, s);
  (void)Fo;
  return 1;
}

int dw_plane_wgrad_try(const float* dz, const float* x, float* dw, int B, int C, int F, int T, int Fo, int To, int k,
                       int stride, int per_plane, const float* in_a, const float* in_b, int in_act, hipStream_t s) {
  if ((long long)B * C > 0x3fffffffLL) return 1;
  if (T > 128 && (long long)F * T < (1 << 28)) {
    TileWgArgs ta{dz, x, dw, B, C, F, T, Fo, To, 0, 0, 0, 1, per_plane, InTf{in_a, in_b, in_act}};
    if (k == 3 && stride == 1) return launch_tile_wgrad<3, 1, 16>(ta, s);
    if (k == 3 && stride == 2) return launch_tile_wgrad<3, 2, 8>(ta, s);
    if (k == 5 && stride == 2) return launch_tile_wgrad<5, 2, 8>(ta, s);
  }
  PlaneWgArgs a{dz, x, dw, B, C, T, To, 1, per_plane, InTf{in_a, in_b, in_act}};
  if (k == 3 && stride == 1 && F == 8 && T > 32 && T <= 64) return launch_plane_wgrad<3, 1, 1, 64, 8>(a, s);
  if (k == 5 && stride == 1 && F == 16 && T > 64 && T <= 128) return launch_plane_wgrad<5, 1, 2, 64, 16>(a, s);
  if (k == 5 && stride == 2 && F == 8 && T > 32 && T <= 64) return launch_plane_wgrad<5, 2, 2, 32, 8>(a, s);
  if (k == 3 && stride == 2 && F == 16 && T > 64 && T <= 128) return launch_plane_wgrad<3, 2, 2, 64, 16>(a, s);
  if (k == 5 && stride == 1 && F == 4 && T <= 32) return launch_plane_wgrad<5, 1, 1, 32, 4>(a, s);
  (void)Fo;
  return 1;
}

int dw_bwd_try(const float* dz, const float* x, const float* in_a, const float* in_b, int in_act, const float* w, float* g,
               float* dw, float* gpart, int* h_inner, int B, int C, int F, int T, int Fo, int To, int k, int stride,
               hipStream_t s, const DwBnBwd* bn, int per_plane_w, const float* res, float* gzpart, int b16) {
  if ((long long)B * C > 0x3fffffffLL || (long long)F * T >= (1 << 28)) return 1;
  // Measured (MI355X, B = 256): the merged kernel wins on the LARGE planes (64x500 -> 32x250: 1.03 vs 1.47 ms, 32x250:
  // 0.35 vs 0.53 ms) and loses on the small late-layer planes, where the whole-plane kernels pack one or two planes per
  // wave with every lane busy (4x32 planes: 0.69 vs 0.35 ms; a wave of this kernel would use 18 of its 64 lanes)
  if (!bn && T <= 128) return 1;                       // (with the BatchNorm backward on load the small planes gain: fewer passes)
  DwBwdArgs a{dz, x, g, dw, gpart, B, C, F, T, Fo, To, 0, 0, 0, 1, InTf{in_a, in_b, in_act},
              DzBn{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, 0, 0}, res, gzpart, b16};
  const bool ppw = per_plane_w != 0;
  if (ppw && !bn) return 1;
  if (bn) a.bn = DzBn{bn->z, bn->a, bn->b, bn->mean, bn->invstd, bn->gscale, bn->gadd, bn->sums, (double)B * Fo * To, bn->act, bn->frozen};
  if (stride == 1 && (Fo != F || To != T)) return 1;
  // 5x5 on planes of <= 4 (output) rows, the last stage of the network: tiles of 4 rows (half the multiply work of RO = 8)
  if (bn && k == 5 && stride == 1 && F <= 4 && T <= 64) return launch_dw_bwd<5, 1, 4>(a, w, h_inner, s, ppw);
  if (bn && k == 5 && stride == 2 && Fo <= 4 && T <= 64) return launch_dw_bwd<5, 2, 4>(a, w, h_inner, s, ppw);
  if (k == 3 && stride == 1) return launch_dw_bwd<3, 1, 8>(a, w, h_inner, s, ppw);      // (RO = 16 needs 246 VGPRs)
  if (k == 5 && stride == 1) return launch_dw_bwd<5, 1, 8>(a, w, h_inner, s, ppw);
  if (k == 3 && stride == 2) return launch_dw_bwd<3, 2, 8>(a, w, h_inner, s, ppw);
  if (k == 5 && stride == 2) return launch_dw_bwd<5, 2, 8>(a, w, h_inner, s, ppw);
  return 1;
}

int dw_tile_dgrad2_try(const float* dz, const float* w, const float* res, float* dx, int B, int C, int F, int T, int Fo,
                       int To, int k, int per_plane_w, hipStream_t s, const DwEpi* epi_) {
  if ((long long)B * C > 0x3fffffffLL || (long long)F * T >= (1 << 28)) return 1;
  const DwEpi epi = epi_ ? *epi_ : DwEpi{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  if (epi.gz && (res || per_plane_w)) return 1;
  TileDgArgs a{dz, res, dx, B, C, F, T, Fo, To, 0, 0, 0, per_plane_w, epi};
  if (k == 3) return launch_tile_dgrad2<3>(a, w, s);
  if (k == 5) return launch_tile_dgrad2<5>(a, w, s);
  return 1;
}

}  // namespace eat

// Partial slots per plane of the merged backward kernel (eat_dw_conv_bwd_g)
extern "C" int eat_dw_bwd_partials_inner(int F, int T, int Fo, int To, int k, int stride) {
  const int wmax = stride == 1 ? (k == 3 ? 125 : 124) : 62;
  const int n_cols = stride == 1 ? T : To, n_rows = stride == 1 ? F : Fo;
  const int ro = 8;
  return ((n_cols + wmax - 1) / wmax) * ((n_rows + ro - 1) / ro);
}

// Upper bound of the partial slots per plane the training epilogues write (the host sizes its buffers with it):
// dgrad == 0: forward conv (F,T) -> (Fo,To); dgrad == 1: data gradient of that conv (dz (Fo,To) -> dx (F,T)).
extern "C" int eat_dw_partials_inner(int F, int T, int Fo, int To, int k, int stride, int dgrad) {
  (void)F;
  if (dgrad && stride == 2) return ((Fo + 7) / 8) * ((To + 61) / 62);           // launch_tile_dgrad2
  const int t_in = dgrad ? T : T, t_out = dgrad ? T : To, f_out = dgrad ? F : Fo, s = dgrad ? 1 : stride;
  if (t_in <= 128) return 1;                                                      // whole-plane kernels / fallback
  const int wmax = s == 1 ? (k == 3 ? 125 : 124) : (k == 3 ? 63 : 62), ro = s == 1 ? 16 : 8;
  return ((t_out + wmax - 1) / wmax) * ((f_out + ro - 1) / ro);
}

