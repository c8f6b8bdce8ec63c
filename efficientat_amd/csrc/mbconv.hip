// Whole inverted-residual block in one kernel for gfx950:
//   expand 1x1 + BN + act  ->  depthwise k x k + BN + act  [-> SE squeeze sums]  [-> project 1x1 + BN (+ residual)]
//   reference: InvertedResidual.forward / its three ConvNormActivation stages
//   (models/mn/block_types.py:138-181) and the SE mean (:72-73).
//
// Why: the expanded tensor (C_exp = 3-6 x C_in channels) is the largest activation of every block.
// Unfused it is written by the expand conv, read and re-written by the depthwise conv and read again by
// the project conv: ~85 % of the block's HBM traffic.  Here it never leaves the CU:
//
//   block = 8 waves on (sample b, TFo x TTo output tile), all channels
//   0. the C_in x (TFi x TTi) input patch (tile + halo) goes ONCE into LDS by per-lane LDS-DMA
//      (global_load_lds_dword gathers the short unaligned patch rows; no staging registers, every row
//      of every channel in flight at once)
//   for every chunk of 16 expanded channels:
//   1. fp32 MFMA (16x16x4, accumulators start at the bias, LDS fragment reads software-pipelined one
//      k-step ahead) -> 16 x P expanded patch in registers; activation, positions outside the image
//      forced to 0 (the depthwise conv zero-pads the ACTIVATED map), written to LDS
//   2. depthwise conv on the LDS patch: thread = (channel, row group, tile column); the whole input
//      strip of the thread is read into registers first (independent LDS reads), then TFR outputs
//   3a. (blocks with SE) outputs go to HBM, plane sums to the SE accumulator - the project conv needs
//       the global squeeze first and stays a separate kernel;
//   3b. (blocks without SE) the 16 x NPo depthwise outputs go to LDS and are multiplied straight into
//       the project conv's accumulators (K = C_exp walked 16 channels per chunk); after the last chunk
//       bias (+ residual) is added and only the C_out x tile result is stored.
//
// The halo is recomputed ((TFi*TTi)/(TFo*TTo*s*s) - 1 extra expand flops: cheap, K = C_in <= 40), so this
// is for the early, bandwidth-bound blocks; the late blocks (8x63 / 4x32 planes, C_in >= 80) are
// MFMA-bound and keep the separate kernels.
#include "eat_common.h"
#include "pw_epilogue.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void;
constexpr int kNT = 512;       // threads per block (8 waves)
constexpr int kNW = kNT / 64;
constexpr int kCC = 16;        // expanded channels per chunk (one MFMA m-tile)
constexpr int kMaxGPW = 2;     // 64-position groups per wave (patch <= 1024 positions)
constexpr int kMaxPPW = 1;     // (project m-tile, 64-output group) pairs per wave

struct MbArgs {
  const float *x, *wpe, *bias_e, *wd, *bias_d, *wpp, *bias_p, *res;
  float *y, *pool;
  int Cin, Cexp, Cout, F, T, Fo, To;
  int MT, MTo;          // m-tiles of the expand / project weights
  int TTo;              // output tile columns (16 or 32); rows = TFR * (kNT / (16 * TTo))
  int tiles_t, Ppad, NPoPad;
  unsigned inv_tti;     // ceil(2^20 / TTi): pos / TTi == (pos * inv_tti) >> 20 for pos < 1024
};

__device__ __forceinline__ void glds4_gather(const float* g, float* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(g) : "memory", "m0");
}

// TFR = output rows per depthwise thread
template <int K, int STRIDE, int TFR, int ACT, bool PROJ>
__global__ __launch_bounds__(kNT, 4) void mbconv_kernel(const MbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int P_ = (K - 1) / 2;
  constexpr int KK = K * K;
  constexpr int SR = (TFR - 1) * STRIDE + K;           // input rows under one thread's outputs
  const int TTo = a.TTo, Cin = a.Cin, Cexp = a.Cexp, F = a.F, T = a.T, Fo = a.Fo, To = a.To;
  const int lg = TTo == 32 ? 5 : 4;
  const int RS = kNT >> (4 + lg);                       // row groups: 1 (TTo 32) or 2 (TTo 16)
  const int TFo = TFR * RS;
  const int TFi = (TFo - 1) * STRIDE + K, TTi = (TTo - 1) * STRIDE + K;
  const int P = TFi * TTi, Ppad = a.Ppad, NG = Ppad >> 6;
  const int NPo = TFo * TTo, NPoPad = a.NPoPad;
  const int nks = Cin >> 2;
  float* Xs = smem;                          // [Cin][Ppad]   input patch
  float* Es = Xs + Cin * Ppad;               // [16][Ppad]    expanded patch of the chunk (later Ds [16][NPoPad])
  float* As = Es + kCC * Ppad;               // [nks][64]     expand A fragments of the chunk
  float* Ws = As + nks * 64;                 // [16][KK+1]    depthwise taps + bias of the chunk
  float* Ap = Ws + kCC * (KK + 1);           // [4][MTo][64]  project A fragments of the chunk
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kq = lane >> 4;
  const int tile = blockIdx.x, b = blockIdx.y;
  const int tf = tile / a.tiles_t, tt = tile - tf * a.tiles_t;
  const int fo0 = tf * TFo, to0 = tt * TTo;
  const int fi0 = fo0 * STRIDE - P_, ti0 = to0 * STRIDE - P_;
  const bool interior = fi0 >= 0 && fi0 + TFi <= F && ti0 >= 0 && ti0 + TTi <= T;
  const size_t plane_in = (size_t)F * T;
  const float* xb = a.x + (size_t)b * Cin * plane_in;

  // ---- weights of one chunk: fetched into registers early, parked in LDS at the point where the
  //      previous readers of that LDS region are known to be done
  const int n_as = nks * 64, n_ws = kCC * (KK + 1);
  float r_as[2], r_ws, r_ap[4];
  auto fetch_as = [&](int mt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + kNT * i;
      r_as[i] = e < n_as ? a.wpe[((size_t)(e >> 6) * a.MT + mt) * 64 + (e & 63)] : 0.0f;
    }
  };
  auto park_as = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + kNT * i;
      if (e < n_as) As[e] = r_as[i];
    }
  };
  auto fetch_ws = [&](int mt) {
    float v = 0.0f;
    if (tid < n_ws) {
      const int ch = tid / (KK + 1), t = tid - ch * (KK + 1);
      const int cg = mt * kCC + ch;
      if (cg < Cexp) v = t < KK ? a.wd[(size_t)cg * KK + t] : a.bias_d[cg];
    }
    r_ws = v;
  };
  auto park_ws = [&]() {
    if (tid < n_ws) Ws[tid] = r_ws;
  };
  auto fetch_ap = [&](int mt) {
    if constexpr (PROJ) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                 // one k-step (MTo fragments of 64 floats) per pass
        const int kg = mt * 4 + ks;
        r_ap[ks] = (tid < a.MTo * 64 && kg * 4 < Cexp) ? a.wpp[(size_t)kg * a.MTo * 64 + tid] : 0.0f;
      }
    }
  };
  auto park_ap = [&]() {
    if constexpr (PROJ) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (tid < a.MTo * 64) Ap[ks * a.MTo * 64 + tid] = r_ap[ks];
    }
  };

  // ---- phase 0: the input patch by LDS-DMA; positions outside the image / beyond the patch read a valid
  //      dummy address: what they hold is never used (E is masked below, pad positions are never read)
#pragma unroll
  for (int q = 0; q < kMaxGPW; ++q) {
    const int g = wv + kNW * q;
    if (g < NG) {
      const int pos = 64 * g + lane;
      int off = 0;
      if (pos < P) {
        const int pi = (int)(((unsigned)pos * a.inv_tti) >> 20), pj = pos - pi * TTi;   // pos / TTi, exact for pos < 1024
        const int fi = fi0 + pi, ti = ti0 + pj;
        if (fi >= 0 && fi < F && ti >= 0 && ti < T) off = fi * T + ti;
      }
      const float* src = xb + off;
      float* dst = Xs + 64 * g;
      for (int k = 0; k < Cin; ++k) glds4_gather(src + (size_t)k * plane_in, dst + k * Ppad);
    }
  }
  fetch_as(0);
  fetch_ws(0);
  fetch_ap(0);
  park_as();
  park_ws();
  park_ap();

  // expand role: wave w owns position groups g = w + 8q; lane owns positions 64 g + 4 (lane&15) .. +3
  unsigned okm = 0xffu;                                // bit q*4+j: position inside the image
  if (!interior) {
    okm = 0;
#pragma unroll
    for (int q = 0; q < kMaxGPW; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pos = 64 * (wv + kNW * q) + 4 * (lane & 15) + j;
        const int pi = (int)(((unsigned)pos * a.inv_tti) >> 20), pj = pos - pi * TTi;
        const int fi = fi0 + pi, ti = ti0 + pj;
        if (pos < P && fi >= 0 && fi < F && ti >= 0 && ti < T) okm |= 1u << (q * 4 + j);
      }
  }
  const bool two_groups = wv + kNW < NG;                // wave-uniform
  // depthwise role: thread = (channel, row group, tile column)
  const int d_tl = tid & (TTo - 1), d_rg = (tid >> lg) & (RS - 1), d_ch = tid >> (lg + (RS >> 1));
  const int d_to = to0 + d_tl, d_fl0 = d_rg * TFR;
  // project role: pairs (m-tile, 64-output group) pr = wv + 8 i
  const int NGo = NPoPad >> 6;
  const int n_pairs = PROJ ? a.MTo * NGo : 0;
  f32x4 accp[kMaxPPW][4];
  if constexpr (PROJ) {
#pragma unroll
    for (int i = 0; i < kMaxPPW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) accp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the asm-issued DMA is invisible to the compiler
  __syncthreads();

  for (int mt = 0; mt < a.MT; ++mt) {
    const bool more = mt + 1 < a.MT;
    // ---- 1. expand: E = act(W_e X + b_e)
    f32x4 acc[kMaxGPW][4];
    {
      f32x4 bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cg = mt * kCC + kq * 4 + r;
        bv[r] = cg < Cexp ? a.bias_e[cg] : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < kMaxGPW; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] = bv;
    }
    {
      const float* xw0 = Xs + kq * Ppad + 64 * wv + 4 * (lane & 15);
      const float* xw1 = xw0 + 64 * kNW;
      float av = As[lane];
      float4 x0 = *reinterpret_cast<const float4*>(xw0);
      float4 x1 = two_groups ? *reinterpret_cast<const float4*>(xw1) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int ks = 0; ks < nks; ++ks) {
        const int kn = ks + 1 < nks ? ks + 1 : ks;
        const float an = As[kn * 64 + lane];
        const float4 n0 = *reinterpret_cast<const float4*>(xw0 + kn * 4 * Ppad);
        float4 n1 = x1;
        if (two_groups) n1 = *reinterpret_cast<const float4*>(xw1 + kn * 4 * Ppad);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x0.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x0.y, acc[0][1], 0, 0, 0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x0.z, acc[0][2], 0, 0, 0);
        acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x0.w, acc[0][3], 0, 0, 0);
        if (two_groups) {
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x1.x, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x1.y, acc[1][1], 0, 0, 0);
          acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x1.z, acc[1][2], 0, 0, 0);
          acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x1.w, acc[1][3], 0, 0, 0);
        }
        av = an; x0 = n0; x1 = n1;
      }
    }
    if (more) fetch_as(mt + 1);                         // lands while the patch is written / convolved
#pragma unroll
    for (int q = 0; q < kMaxGPW; ++q) {
      const int g = wv + kNW * q;
      if (g < NG) {
        const int pos0 = 64 * g + 4 * (lane & 15);
        const unsigned m4 = okm >> (q * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float4 v;
          v.x = eat::activate<ACT>(acc[q][0][r]);
          v.y = eat::activate<ACT>(acc[q][1][r]);
          v.z = eat::activate<ACT>(acc[q][2][r]);
          v.w = eat::activate<ACT>(acc[q][3][r]);
          if (!interior) {
            v.x = (m4 & 1u) ? v.x : 0.0f;
            v.y = (m4 & 2u) ? v.y : 0.0f;
            v.z = (m4 & 4u) ? v.z : 0.0f;
            v.w = (m4 & 8u) ? v.w : 0.0f;
          }
          *reinterpret_cast<float4*>(Es + (kq * 4 + r) * Ppad + pos0) = v;
        }
      }
    }
    __syncthreads();                                    // S1: E visible; everyone is done with As
    if (more) park_as();
    if (more) fetch_ws(mt + 1);
    if (PROJ && more) fetch_ap(mt + 1);

    // ---- 2. depthwise on the LDS patch
    float dres[TFR];
    {
      const int cg = mt * kCC + d_ch;
      const float* ep = Es + d_ch * Ppad + (d_fl0 * STRIDE) * TTi + d_tl * STRIDE;
      float col[SR][K];
#pragma unroll
      for (int u = 0; u < SR; ++u)
#pragma unroll
        for (int v = 0; v < K; ++v) col[u][v] = ep[u * TTi + v];
      float wr[KK];
#pragma unroll
      for (int i = 0; i < KK; ++i) wr[i] = Ws[d_ch * (KK + 1) + i];
      const float bd = Ws[d_ch * (KK + 1) + KK];
      const bool col_ok = cg < Cexp && d_to < To;
      float* yb = PROJ ? nullptr : a.y + (((size_t)b * Cexp + cg) * Fo + fo0 + d_fl0) * To + d_to;
      float psum = 0.0f;
#pragma unroll
      for (int fl = 0; fl < TFR; ++fl) {
        float sacc = bd;
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
          for (int v = 0; v < K; ++v) sacc = fmaf(wr[u * K + v], col[fl * STRIDE + u][v], sacc);
        float ov = eat::activate<ACT>(sacc);
        const bool ok = col_ok && fo0 + d_fl0 + fl < Fo;
        ov = ok ? ov : 0.0f;
        if constexpr (!PROJ) {
          if (ok) yb[(size_t)fl * To] = ov;
          psum += ov;
        }
        dres[fl] = ov;
      }
      if constexpr (!PROJ) {
        if (a.pool) {     // the TTo*RS consecutive lanes of one channel: one atomic per channel per tile
          const int span = TTo * RS;                    // 32
          for (int o = span >> 1; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
          if ((tid & (span - 1)) == 0 && cg < Cexp) atomicAdd(a.pool + (size_t)b * Cexp + cg, psum);
        }
      }
    }
    __syncthreads();                                    // S2: everyone is done with Es and Ws
    if (more) park_ws();
    if constexpr (PROJ) {
      // ---- 3b. D (16 x NPo) -> LDS (over the expanded patch), then into the project accumulators
      float* Ds = Es;
#pragma unroll
      for (int fl = 0; fl < TFR; ++fl) Ds[d_ch * NPoPad + (d_fl0 + fl) * TTo + d_tl] = dres[fl];
      __syncthreads();                                  // S3: D visible
      const int ksn = (Cexp - mt * kCC) >= kCC ? 4 : ((Cexp - mt * kCC) >> 2);
#pragma unroll
      for (int i = 0; i < kMaxPPW; ++i) {
        const int pr = wv + kNW * i;
        if (pr < n_pairs) {
          const int m = pr / NGo, g = pr - m * NGo;
          float avp[4];
          float4 dv[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int kc = ks < ksn ? ks : 0;
            avp[ks] = Ap[(kc * a.MTo + m) * 64 + lane];
            dv[ks] = *reinterpret_cast<const float4*>(Ds + (kc * 4 + kq) * NPoPad + 64 * g + 4 * (lane & 15));
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            if (ks < ksn) {
              accp[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(avp[ks], dv[ks].x, accp[i][0], 0, 0, 0);
              accp[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(avp[ks], dv[ks].y, accp[i][1], 0, 0, 0);
              accp[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(avp[ks], dv[ks].z, accp[i][2], 0, 0, 0);
              accp[i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(avp[ks], dv[ks].w, accp[i][3], 0, 0, 0);
            }
          }
        }
      }
      __syncthreads();                                  // S4: everyone is done with Ds and Ap
      if (more) park_ap();
    }
    // hazards: As is parked after S1 and read after S2 (next chunk's MFMA loop); Ws is parked after S2 and
    // read after the next S1; Ap is parked after S4 and read after the next S3; the next chunk's E write
    // comes after S2 (S4 with the project stage), i.e. after the last reader of Es / Ds.
  }

  if constexpr (PROJ) {
    // ---- project epilogue: accumulators -> LDS -> coalesced rows with bias (+ residual)
    float* Os = smem;                                   // [MTo*16][NPoPad] over the patch buffers
#pragma unroll
    for (int i = 0; i < kMaxPPW; ++i) {
      const int pr = wv + kNW * i;
      if (pr < n_pairs) {
        const int m = pr / NGo, g = pr - m * NGo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 v = make_float4(accp[i][0][r], accp[i][1][r], accp[i][2][r], accp[i][3][r]);
          *reinterpret_cast<float4*>(Os + (m * 16 + kq * 4 + r) * NPoPad + 64 * g + 4 * (lane & 15)) = v;
        }
      }
    }
    __syncthreads();
    const int Cout = a.Cout;
    const int lgn = 31 - __builtin_clz(NPo);
    for (int e = tid; e < Cout * NPo; e += kNT) {
      const int co = e >> lgn, p = e & (NPo - 1);        // NPo is a power of two
      const int fl = p >> lg, tl = p & (TTo - 1);
      const int fo = fo0 + fl, to = to0 + tl;
      if (fo < Fo && to < To) {
        const size_t o = (((size_t)b * Cout + co) * Fo + fo) * To + to;
        float v = Os[co * NPoPad + p] + a.bias_p[co];
        if (a.res) v += a.res[o];
        a.y[o] = v;
      }
    }
  }
}

template <int K, int STRIDE, int TFR, int ACT, bool PROJ>
int launch_one(const MbArgs& a, dim3 grid, size_t smem, hipStream_t s) {
  auto kern = mbconv_kernel<K, STRIDE, TFR, ACT, PROJ>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eat::fail(EAT_ELAUNCH, "eat_mbconv_fwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, grid, dim3(kNT), smem, s, a);
  return eat::check_launch("eat_mbconv_fwd");
}

template <int K, int STRIDE>
int launch_mb(const MbArgs& in, int B, int act, hipStream_t s) {
  MbArgs a = in;
  const bool proj = a.wpp != nullptr;
  a.MT = (a.Cexp + 15) / 16;
  a.MTo = proj ? (a.Cout + 15) / 16 : 0;
  auto ppad_of = [&](int tfo, int tto) { return (((tfo - 1) * STRIDE + K) * ((tto - 1) * STRIDE + K) + 63) & ~63; };
  auto lds_floats = [&](int tfo, int tto) {
    return (size_t)(a.Cin + kCC) * ppad_of(tfo, tto) + (a.Cin / 4) * 64 + kCC * (K * K + 1) + (proj ? 4 * a.MTo * 64 : 0);
  };
  // candidate tiles (rows, cols, rows per depthwise thread), largest first; the patch must fit 1024
  // positions and the LDS must let two blocks share a CU (<= 78 KB each)
  const int cand[][3] = {{8, 32, 8}, {4, 32, 4}, {8, 16, 4}, {4, 16, 2}, {2, 16, 1}};
  int TFo = 0, TTo = 0, TFR = 0;
  for (const auto& c : cand) {
    if (c[0] > a.Fo && c[0] > 2) continue;               // do not waste rows on short planes
    if (K == 5 && c[2] == 8) continue;                   // 12-19 x 5 strip per thread: does not fit 128 VGPRs
    const int ppad = ppad_of(c[0], c[1]);
    if (ppad > 64 * kNW * kMaxGPW) continue;
    if (lds_floats(c[0], c[1]) * sizeof(float) > 78 * 1024) continue;
    if (proj) {
      const int npad = (c[0] * c[1] + 63) & ~63;
      if (a.MTo * (npad >> 6) > kNW * kMaxPPW) continue;                     // project pairs per wave
      if ((size_t)a.MTo * 16 * npad > (size_t)(a.Cin + kCC) * ppad) continue;   // Os fits over the patches
      if (npad > ppad) continue;                                             // Ds fits over Es
    }
    TFo = c[0]; TTo = c[1]; TFR = c[2];
    break;
  }
  if (!TFo) return eat::fail(EAT_EINVAL, "eat_mbconv_fwd: no tile fits (Cin=%d Cexp=%d Cout=%d k=%d s=%d)", a.Cin, a.Cexp, a.Cout, K, STRIDE);
  a.TTo = TTo;
  a.Ppad = ppad_of(TFo, TTo);
  a.NPoPad = (TFo * TTo + 63) & ~63;
  a.tiles_t = (a.To + TTo - 1) / TTo;
  {
    const unsigned tti = (unsigned)((TTo - 1) * STRIDE + K);
    a.inv_tti = ((1u << 20) + tti - 1) / tti;
  }
  const int tiles_f = (a.Fo + TFo - 1) / TFo;
  const size_t smem = lds_floats(TFo, TTo) * sizeof(float);
  dim3 grid(a.tiles_t * tiles_f, B);
#define EAT_MB_CASE(R)                                                                      \
  case R:                                                                                   \
    if (act == EAT_ACT_RELU)                                                                \
      return proj ? launch_one<K, STRIDE, R, EAT_ACT_RELU, true>(a, grid, smem, s)          \
                  : launch_one<K, STRIDE, R, EAT_ACT_RELU, false>(a, grid, smem, s);        \
    return proj ? launch_one<K, STRIDE, R, EAT_ACT_HSWISH, true>(a, grid, smem, s)          \
                : launch_one<K, STRIDE, R, EAT_ACT_HSWISH, false>(a, grid, smem, s);
  switch (TFR) {
    EAT_MB_CASE(1) EAT_MB_CASE(2) EAT_MB_CASE(4) EAT_MB_CASE(8)
  }
#undef EAT_MB_CASE
  return eat::fail(EAT_EINVAL, "eat_mbconv_fwd: internal tiling error");
}

int mbconv(const float* x, const float* wp_e, const float* bias_e, const float* w_d, const float* bias_d,
           const float* wp_p, const float* bias_p, const float* res, float* y, float* pool, int B, int Cin, int Cexp,
           int Cout, int F, int T, int Fo, int To, int k, int stride, int act, hipStream_t s, const char* who) {
  if (Cin % 4 != 0 || Cin > 40) return eat::fail(EAT_EINVAL, "%s: Cin=%d must be a multiple of 4 and <= 40", who, Cin);
  if (Cexp % 4 != 0) return eat::fail(EAT_EINVAL, "%s: Cexp=%d must be a multiple of 4", who, Cexp);
  if (act != EAT_ACT_RELU && act != EAT_ACT_HSWISH) return eat::fail(EAT_EINVAL, "%s: act must be relu/hswish", who);
  if (wp_p && Cout > 96) return eat::fail(EAT_EINVAL, "%s: Cout=%d > 96 unsupported by the fused project stage", who, Cout);
  const int p = (k - 1) / 2;
  if (Fo != (F + 2 * p - k) / stride + 1 || To != (T + 2 * p - k) / stride + 1)
    return eat::fail(EAT_EINVAL, "%s: output %dx%d inconsistent with input %dx%d", who, Fo, To, F, T);
  {   // round 2: the register-resident kernel (irb.hip) where it has an instantiation; this LDS-staged one otherwise
    const int rc = eat::irb_try(x, wp_e, bias_e, w_d, bias_d, wp_p, bias_p, res, y, pool, B, Cin, Cexp, Cout, F, T, Fo, To,
                                k, stride, act, s);
    if (rc != 1) return rc;
  }
  MbArgs a{};
  a.x = x; a.wpe = wp_e; a.bias_e = bias_e; a.wd = w_d; a.bias_d = bias_d; a.wpp = wp_p; a.bias_p = bias_p; a.res = res;
  a.y = y; a.pool = pool;
  a.Cin = Cin; a.Cexp = Cexp; a.Cout = Cout; a.F = F; a.T = T; a.Fo = Fo; a.To = To;
  if (k == 3 && stride == 1) return launch_mb<3, 1>(a, B, act, s);
  if (k == 3 && stride == 2) return launch_mb<3, 2>(a, B, act, s);
  if (k == 5 && stride == 1) return launch_mb<5, 1>(a, B, act, s);
  if (k == 5 && stride == 2) return launch_mb<5, 2>(a, B, act, s);
  return eat::fail(EAT_EINVAL, "%s: unsupported k=%d stride=%d", who, k, stride);
}

}  // namespace

extern "C" int eat_fused_expand_dw_fwd(const float* x, const float* wp_e, const float* bias_e, const float* w_d,
                                       const float* bias_d, float* y, float* pool, int B, int Cin, int Cexp, int F,
                                       int T, int Fo, int To, int k, int stride, int act, eat_stream_t stream) {
  eat::clear_stale_error();
  return mbconv(x, wp_e, bias_e, w_d, bias_d, nullptr, nullptr, nullptr, y, pool, B, Cin, Cexp, 0, F, T, Fo, To, k, stride,
                act, (hipStream_t)stream, "eat_fused_expand_dw_fwd");
}

extern "C" int eat_mbconv_fwd(const float* x, const float* wp_e, const float* bias_e, const float* w_d,
                              const float* bias_d, const float* wp_p, const float* bias_p, const float* res, float* y,
                              int B, int Cin, int Cexp, int Cout, int F, int T, int Fo, int To, int k, int stride,
                              int act, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!wp_p || !bias_p) return eat::fail(EAT_EINVAL, "eat_mbconv_fwd: project weights and bias are required");
  return mbconv(x, wp_e, bias_e, w_d, bias_d, wp_p, bias_p, nullptr == res ? nullptr : res, y, nullptr, B, Cin, Cexp, Cout, F, T,
                Fo, To, k, stride, act, (hipStream_t)stream, "eat_mbconv_fwd");
}
