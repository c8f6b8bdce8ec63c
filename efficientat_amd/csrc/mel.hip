// Fused log-mel front-end for gfx950: pre-emphasis -> reflect-padded hann-windowed frames ->
// 1024-point real FFT -> power -> banded mel filterbank -> log -> (masking) -> normalise.
// Replaces the torch-op chain of models/preprocess.py:40-67 (one launch instead of ~8 plus a
// host-built basis upload per call).
//
// Mapping: one wavefront (64 lanes) per frame.  The 1024 real samples are packed as 512 complex
// points; three radix-8 Stockham passes (8 points per lane, in registers) exchange through a
// private LDS buffer of the wave, so the FFT needs no workgroup barrier.  A block of 4 waves
// produces 4 tiles of 16 consecutive frames and stages each (n_mels x 16) output tile in LDS so the
// store to out (B, n_mels, T) is coalesced along the time axis.
//
// Round 2 (the round-1 kernel was LDS-bound: 820 LDS cycles per frame per CU, 42 % of them bank
// conflicts, against ~350 VALU cycles):
//   * the twiddles of the two twiddled passes are read from COMPACT per-pass LDS tables laid out in
//     lane order ([r][lane & 7] and [r][lane]: conflict-free 8-byte reads) and the real-FFT unpack
//     twiddles are one register (w^lane) times compile-time constants - the round-1 kernel indexed
//     the full 1024-entry table with strides of 16 r k and 2 r lane entries: 4- to 8-way conflicts;
//   * the exchange buffer is XOR-swizzled (float2 index i -> i ^ ((i >> 3) & 15)), conflict-free
//     for the pass-1 / pass-2 writes (16-lane ds_write_b64 groups) AND the strided reads
//     (32-lane ds_read_b64 groups) - no padding;
//   * the real-FFT unpack needs Z[k] and Z[512-k]: after the last pass these sit in lanes l and
//     64-l, so they are exchanged with 16 ds_bpermute instead of a third LDS round trip;
//   * the filterbank reads the power spectrum and the band weights as 8-byte pairs (band starts
//     aligned to even bins) and walks only as many pairs as the widest band of the lane group
//     needs (low mel filters are 2-6 bins wide, the top ones 24): ~35 instead of 96 LDS reads.
#include "eat_common.h"

namespace {

constexpr int kNfft = 1024;
constexpr int kHalf = 512;            // complex points
constexpr int kFramesPerBlock = 16;   // frames of one output tile (staged in LDS, stored coalesced)
constexpr int kGroupsPerBlock = 4;    // tiles per block: the tables are loaded once per 64 frames
constexpr int kWavesPerBlock = 4;
constexpr int kMaxRounds = 4;         // n_mels <= 256

__device__ __forceinline__ int swz(int i) { return i ^ ((i >> 3) & 15); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

// In-register 8-point DFT (decimation in frequency); result in natural order in u[].
__device__ __forceinline__ void dft8(float2 (&u)[8]) {
  constexpr float h = 0.70710678118654752440f;
  float2 a0 = cadd(u[0], u[4]), a4 = csub(u[0], u[4]);
  float2 a1 = cadd(u[1], u[5]), d1 = csub(u[1], u[5]);
  float2 a2 = cadd(u[2], u[6]), a6 = mul_mi(csub(u[2], u[6]));
  float2 a3 = cadd(u[3], u[7]), d3 = csub(u[3], u[7]);
  float2 a5 = make_float2(h * (d1.x + d1.y), h * (d1.y - d1.x));    // * (1 - i)/sqrt2
  float2 a7 = make_float2(h * (d3.y - d3.x), -h * (d3.x + d3.y));   // * (-1 - i)/sqrt2
  float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3));
  float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = mul_mi(csub(a5, a7));
  u[0] = cadd(b0, b1); u[4] = csub(b0, b1); u[2] = cadd(b2, b3); u[6] = csub(b2, b3);
  u[1] = cadd(b4, b5); u[5] = csub(b4, b5); u[3] = cadd(b6, b7); u[7] = csub(b6, b7);
}

// Orders this wave's LDS traffic (the buffer is private to the wave: no s_barrier needed).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float lane_read(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

__global__ __launch_bounds__(256) void mel_fwd_kernel(
    const float* __restrict__ wave, int L, const float* __restrict__ window, int win_length, int hop,
    const float2* __restrict__ twiddle, const float2* __restrict__ band_w2,
    const int* __restrict__ band_start, const int* __restrict__ band_cnt, int n_mels, int band_pairs,
    float* __restrict__ out, int T, int mf0, int mf1, int mt0, int mt1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* s_buf = reinterpret_cast<float2*>(smem);                         // [4][512]  swizzled exchange buffers
  float2* s_bw = s_buf + kWavesPerBlock * kHalf;                           // [band_pairs][n_mels] weight pairs
  float* s_win = reinterpret_cast<float*>(s_bw + band_pairs * n_mels);     // [1024] zero-padded window
  float2* s_tw2 = reinterpret_cast<float2*>(s_win + kNfft);                // [7][8]   w_64^(r k), k = lane & 7
  float2* s_tw3 = s_tw2 + 7 * 8;                                           // [7][64]  w_512^(r lane)
  int* s_bs = reinterpret_cast<int*>(s_tw3 + 7 * 64);                      // [n_mels] even band starts
  float* s_out = reinterpret_cast<float*>(s_bs + n_mels);                  // [n_mels][17]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  const int lpad = (kNfft - win_length) / 2;

  for (int i = tid; i < kNfft; i += 256) {
    const int wi = i - lpad;
    s_win[i] = (wi >= 0 && wi < win_length) ? window[wi] : 0.0f;
  }
  for (int i = tid; i < band_pairs * n_mels; i += 256) s_bw[i] = band_w2[i];
  for (int i = tid; i < n_mels; i += 256) s_bs[i] = band_start[i];

  for (int i = tid; i < 7 * 8; i += 256) s_tw2[i] = twiddle[16 * (i / 8 + 1) * (i & 7)];
  for (int i = tid; i < 7 * 64; i += 256) s_tw3[i] = twiddle[2 * (i / 64 + 1) * (i & 63)];
  const float2 twu0 = twiddle[lane];                 // unpack: w_1024^(lane + 64 m) = twu0 * w_16^m
  // filterbank: pairs to walk per 64-mel round = widest band of the round (wave-uniform)
  auto round_pairs = [&](int rd) {
    const int m = lane + 64 * rd;
    int c = m < n_mels ? band_cnt[m] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o, 64));
    return __builtin_amdgcn_readfirstlane(c);
  };
  const int np0 = round_pairs(0), np1 = round_pairs(1), np2 = round_pairs(2), np3 = round_pairs(3);
  __syncthreads();

  const float* x = wave + (size_t)b * L;
  const int Lp = L - 1;  // length of the pre-emphasised signal
  float2* buf = s_buf + wv * kHalf;
  float* pw = reinterpret_cast<float*>(buf);     // power spectrum aliases the wave's FFT buffer

  // Raw samples of one frame: lane holds frame positions n = 2*(lane + 64 r) + e.  Interior frames (no reflection):
  // (x[j], x[j+1]) as one aligned 8-byte load and x[j+2] as a third word - raw[r] = (x0, x1, x2), pre-emphasised later.
  // Edge frames (first / last two of a clip): raw[r] = (pre[n], pre[n+1], -) computed here.  All loads are
  // unconditional so they issue back to back; the next frame's loads are issued before the current frame's FFT.
  auto load_frame = [&](int t, float (&raw)[8][3]) -> bool {
    const int q0 = t * hop - kNfft / 2;        // padded-signal origin of this frame, in pre[] indices
    if (q0 >= 0 && q0 + kNfft <= Lp && ((q0 | L) & 1) == 0) {
      const float* xf = x + q0 + 2 * lane;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float2 a = *reinterpret_cast<const float2*>(xf + 128 * r);
        raw[r][0] = a.x; raw[r][1] = a.y; raw[r][2] = xf[128 * r + 2];
      }
      return false;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float pv[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int j = q0 + 2 * (lane + 64 * r) + e;
        j = j < 0 ? -j : j;
        j = j >= Lp ? 2 * (Lp - 1) - j : j;
        j = j < 0 ? 0 : (j > Lp - 1 ? Lp - 1 : j);   // only reachable where the window is zero
        pv[e] = __fsub_rn(x[j + 1], __fmul_rn(0.97f, x[j]));
      }
      raw[r][0] = pv[0]; raw[r][1] = pv[1]; raw[r][2] = 0.0f;
    }
    return true;
  };
  float raw[8][3];                              // ONE staging buffer: converted at the top of a frame, then refilled
  bool raw_pre = false;
  float* o = out + (size_t)b * n_mels * T;
  for (int grp = 0; grp < kGroupsPerBlock; ++grp) {
  const int t_base = (blockIdx.x * kGroupsPerBlock + grp) * kFramesPerBlock;
  if (t_base >= T) break;                      // block-uniform
  if (t_base + wv < T) raw_pre = load_frame(t_base + wv, raw);

  for (int fi = 0; fi < kFramesPerBlock / kWavesPerBlock; ++fi) {
    const int fl = fi * kWavesPerBlock + wv;   // frame slot inside the block tile
    const int t = t_base + fl;
    if (t < T) {                               // wave-uniform
      const bool have_next = (fi + 1 < kFramesPerBlock / kWavesPerBlock) && (t + kWavesPerBlock < T);
      float2 u[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        // pre[j] = x[j+1] - 0.97*x[j]   (conv1d with [-0.97, 1], preprocess.py:30,41), then window
        const float2 w = *reinterpret_cast<const float2*>(s_win + 2 * (lane + 64 * r));
        const float p0 = raw_pre ? raw[r][0] : __fsub_rn(raw[r][1], __fmul_rn(0.97f, raw[r][0]));
        const float p1 = raw_pre ? raw[r][1] : __fsub_rn(raw[r][2], __fmul_rn(0.97f, raw[r][1]));
        u[r] = make_float2(__fmul_rn(w.x, p0), __fmul_rn(w.y, p1));
      }
      // the staging registers are free again: the next frame's loads fly during this frame's FFT
      if (have_next) raw_pre = load_frame(t + kWavesPerBlock, raw);
      // pass 1: p = 1 (no twiddles)
      dft8(u);
#pragma unroll
      for (int s = 0; s < 8; ++s) buf[swz(8 * lane + s)] = u[s];
      wave_lds_fence();
      // pass 2: p = 8
      {
        const int k = lane & 7;
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = buf[swz(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], s_tw2[(r - 1) * 8 + k]);
        dft8(u);
        wave_lds_fence();
        const int j = (lane - k) * 8 + k;
#pragma unroll
        for (int s = 0; s < 8; ++s) buf[swz(j + 8 * s)] = u[s];
        wave_lds_fence();
      }
      // pass 3: p = 64; the result Z[lane + 64 s] stays in u[s]
      {
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = buf[swz(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], s_tw3[(r - 1) * 64 + lane]);
        dft8(u);
      }
      // unpack the real FFT and take the power:  X[k] = E[k] + w^k O[k], k = lane + 64 m.
      // Z[512 - k] is u[7 - m] of lane 64 - lane (lane 0: its own u[(8 - m) & 7]).
      float pk[8];
      {
        const int src = (64 - lane) & 63;
        constexpr float c16[8] = {1.0f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f,
                                  0.0f, -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f};
        constexpr float s16[8] = {0.0f, -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f,
                                  -1.0f, -0.92387953251128675613f, -0.70710678118654752440f, -0.38268343236508977173f};
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float2 zk = u[m];
          float2 zn = make_float2(lane_read(u[7 - m].x, src), lane_read(u[7 - m].y, src));
          if (lane == 0) zn = u[(8 - m) & 7];
          const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
          const float2 od = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
          const float2 tw = cmul(twu0, make_float2(c16[m], s16[m]));      // exp(-2 pi i (lane + 64 m) / 1024)
          const float2 xk = cadd(e, cmul(tw, od));
          pk[m] = xk.x * xk.x + xk.y * xk.y;
        }
      }
      wave_lds_fence();                        // every lane is done reading the exchange buffer
#pragma unroll
      for (int m = 0; m < 8; ++m) pw[lane + 64 * m] = pk[m];
      wave_lds_fence();
      // banded mel filterbank + log + masking + normalisation
#pragma unroll
      for (int rd = 0; rd < kMaxRounds; ++rd) {
        const int m = lane + 64 * rd;
        if (64 * rd < n_mels) {                // wave-uniform
          const int mc = m < n_mels ? m : n_mels - 1;
          const float2* pp = reinterpret_cast<const float2*>(pw + s_bs[mc]);
          const float2* wp = s_bw + mc;
          float acc = 0.0f;
          const int np = rd == 0 ? np0 : rd == 1 ? np1 : rd == 2 ? np2 : np3;
          for (int j = 0; j < np; ++j) {
            const float2 w = wp[j * n_mels];
            const float2 p = pp[j];
            acc = fmaf(w.y, p.y, fmaf(w.x, p.x, acc));
          }
          float v = logf(acc + 0.00001f);
          if ((m >= mf0 && m < mf1) || (t >= mt0 && t < mt1)) v = 0.0f;
          if (m < n_mels) s_out[m * (kFramesPerBlock + 1) + fl] = (v + 4.5f) / 5.0f;
        }
      }
      wave_lds_fence();
    }
  }
  __syncthreads();
  for (int i = tid; i < n_mels * kFramesPerBlock; i += 256) {
    const int m = i / kFramesPerBlock, c = i % kFramesPerBlock;
    const int t = t_base + c;
    if (t < T) o[(size_t)m * T + t] = s_out[m * (kFramesPerBlock + 1) + c];
  }
  __syncthreads();                             // s_out is rewritten by the next tile
  }
}

}  // namespace

extern "C" int eat_mel_fwd(const float* wave, int B, int L, const float* window, int win_length,
                           int n_fft, int hop, const float* twiddle, const float* band_w2,
                           const int* band_start, const int* band_cnt, int n_mels, int band_pairs, float* out,
                           int T, int mask_f0, int mask_f1, int mask_t0, int mask_t1, eat_stream_t stream) {
  eat::clear_stale_error();
  if (n_fft != kNfft) return eat::fail(EAT_EINVAL, "eat_mel_fwd: only n_fft=1024 is implemented (got %d)", n_fft);
  if (win_length < 1 || win_length > n_fft || hop < 1 || B < 1 || n_mels < 1 || band_pairs < 1)
    return eat::fail(EAT_EINVAL, "eat_mel_fwd: bad geometry");
  if (n_mels > 64 * kMaxRounds) return eat::fail(EAT_EINVAL, "eat_mel_fwd: n_mels=%d > %d", n_mels, 64 * kMaxRounds);
  if (L - 1 <= n_fft / 2) return eat::fail(EAT_EINVAL, "eat_mel_fwd: clip too short for reflect padding (L=%d)", L);
  if (T != 1 + (L - 1) / hop) return eat::fail(EAT_EINVAL, "eat_mel_fwd: T=%d does not match L=%d hop=%d", T, L, hop);
  size_t smem = sizeof(float2) * (kWavesPerBlock * kHalf + (size_t)band_pairs * n_mels + 7 * 8 + 7 * 64) + sizeof(float) * kNfft +
                sizeof(int) * n_mels + sizeof(float) * (size_t)n_mels * (kFramesPerBlock + 1);
  if (smem > 160 * 1024) return eat::fail(EAT_EINVAL, "eat_mel_fwd: mel table too large for LDS (%zu B)", smem);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mel_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eat::fail(EAT_ELAUNCH, "eat_mel_fwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  dim3 grid((T + kFramesPerBlock * kGroupsPerBlock - 1) / (kFramesPerBlock * kGroupsPerBlock), B);
  hipLaunchKernelGGL(mel_fwd_kernel, grid, dim3(256), smem, (hipStream_t)stream, wave, L, window, win_length,
                     hop, reinterpret_cast<const float2*>(twiddle), reinterpret_cast<const float2*>(band_w2), band_start,
                     band_cnt, n_mels, band_pairs, out, T, mask_f0, mask_f1, mask_t0, mask_t1);
  return eat::check_launch("eat_mel_fwd");
}
