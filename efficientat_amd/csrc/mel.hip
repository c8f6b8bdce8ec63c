// Fused log-mel front-end for gfx950: pre-emphasis -> reflect-padded hann-windowed frames ->
// 1024-point real FFT -> power -> banded mel filterbank -> log -> (masking) -> normalise.
// Replaces the torch-op chain of models/preprocess.py:40-67 (one launch instead of ~8 plus a
// host-built basis upload per call).
//
// Mapping: one wavefront (64 lanes) per frame.  The 1024 real samples are packed as 512
// complex points; three radix-8 Stockham passes (8 points per lane, in registers) exchange
// through a private LDS buffer of the wave, so the FFT needs no workgroup barrier.  A block of
// 4 waves produces 16 consecutive frames and stages its (n_mels x 16) output tile in LDS so
// the store to out (B, n_mels, T) is coalesced along the time axis.
#include "eat_common.h"

namespace {

constexpr int kNfft = 1024;
constexpr int kHalf = 512;        // complex points
constexpr int kFramesPerBlock = 16;   // frames of one output tile (staged in LDS, stored coalesced)
constexpr int kGroupsPerBlock = 4;    // tiles per block: the 28 KB of tables are loaded once per 64 frames
constexpr int kWavesPerBlock = 4;
constexpr int kBufStride = kHalf + kHalf / 8;  // padded: idx + idx/8 (breaks the stride-8 store conflict)

__device__ __forceinline__ int padidx(int i) { return i + (i >> 3); }

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

__global__ __launch_bounds__(256) void mel_fwd_kernel(
    const float* __restrict__ wave, int L, const float* __restrict__ window, int win_length, int hop,
    const float2* __restrict__ twiddle, const float* __restrict__ band_w,
    const int* __restrict__ band_start, int n_mels, int band_len, float* __restrict__ out, int T,
    int mf0, int mf1, int mt0, int mt1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* s_tw = reinterpret_cast<float2*>(smem);                          // [1024]
  float2* s_buf = s_tw + kNfft;                                            // [4][kBufStride]
  float* s_win = reinterpret_cast<float*>(s_buf + kWavesPerBlock * kBufStride);  // [1024] zero-padded window
  float* s_bw = s_win + kNfft;                                             // [band_len][n_mels]
  int* s_bs = reinterpret_cast<int*>(s_bw + band_len * n_mels);            // [n_mels]
  float* s_out = reinterpret_cast<float*>(s_bs + n_mels);                  // [n_mels][33]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  const int lpad = (kNfft - win_length) / 2;

  for (int i = tid; i < kNfft; i += 256) {
    s_tw[i] = twiddle[i];
    int wi = i - lpad;
    s_win[i] = (wi >= 0 && wi < win_length) ? window[wi] : 0.0f;
  }
  for (int i = tid; i < band_len * n_mels; i += 256) {
    int m = i % n_mels, j = i / n_mels;          // transposed: lanes walk mel rows
    s_bw[i] = band_w[m * band_len + j];
  }
  for (int i = tid; i < n_mels; i += 256) s_bs[i] = band_start[i];
  __syncthreads();

  const float* x = wave + (size_t)b * L;
  const int Lp = L - 1;  // length of the pre-emphasised signal
  float2* buf = s_buf + wv * kBufStride;
  float* pw = reinterpret_cast<float*>(buf);     // power spectrum aliases the wave's FFT buffer

  // Raw samples of one frame: lane holds frame positions n = 2*(lane + 64 r) + e as (x[j], x[j+1])
  // with j the reflect-padded index into the pre-emphasised signal.  All 32 loads are
  // unconditional (clamped) so they issue back to back; the next frame's loads are issued before
  // the current frame's FFT so their latency hides behind it.
  auto load_frame = [&](int t, float (&lo)[16], float (&hi)[16]) {
    const int q0 = t * hop - kNfft / 2;        // padded-signal origin of this frame, in pre[] indices
    if (q0 >= 0 && q0 + kNfft <= Lp && ((q0 | L) & 1) == 0) {
      // interior frame (no reflection): x[j], x[j+1] as one aligned 8-byte load, x[j+2] as a third word
      const float* xf = x + q0 + 2 * lane;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float2 a = *reinterpret_cast<const float2*>(xf + 128 * r);
        const float c = xf[128 * r + 2];
        lo[2 * r] = a.x; hi[2 * r] = a.y; lo[2 * r + 1] = a.y; hi[2 * r + 1] = c;
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int j = q0 + 2 * (lane + 64 * r) + e;
        j = j < 0 ? -j : j;
        j = j >= Lp ? 2 * (Lp - 1) - j : j;
        j = j < 0 ? 0 : (j > Lp - 1 ? Lp - 1 : j);   // only reachable where the window is zero
        lo[2 * r + e] = x[j];
        hi[2 * r + e] = x[j + 1];
      }
  };
  float cur_lo[16], cur_hi[16], nxt_lo[16], nxt_hi[16];
  float* o = out + (size_t)b * n_mels * T;
  for (int grp = 0; grp < kGroupsPerBlock; ++grp) {
  const int t_base = (blockIdx.x * kGroupsPerBlock + grp) * kFramesPerBlock;
  if (t_base >= T) break;                      // block-uniform
  if (t_base + wv < T) load_frame(t_base + wv, cur_lo, cur_hi);

  for (int fi = 0; fi < kFramesPerBlock / kWavesPerBlock; ++fi) {
    const int fl = fi * kWavesPerBlock + wv;   // frame slot inside the block tile
    const int t = t_base + fl;
    if (t < T) {                               // wave-uniform
      const bool have_next = (fi + 1 < kFramesPerBlock / kWavesPerBlock) && (t + kWavesPerBlock < T);
      if (have_next) load_frame(t + kWavesPerBlock, nxt_lo, nxt_hi);
      float2 u[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = 2 * (lane + 64 * r);
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          // pre[j] = x[j+1] - 0.97*x[j]   (conv1d with [-0.97, 1], preprocess.py:30,41), then window
          v[e] = __fmul_rn(s_win[n + e], __fsub_rn(cur_hi[2 * r + e], __fmul_rn(0.97f, cur_lo[2 * r + e])));
        }
        u[r] = make_float2(v[0], v[1]);
      }
      // pass 1: p = 1 (no twiddles)
      dft8(u);
#pragma unroll
      for (int s = 0; s < 8; ++s) buf[padidx(8 * lane + s)] = u[s];
      wave_lds_fence();
      // pass 2: p = 8
      {
        const int k = lane & 7;
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = buf[padidx(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], s_tw[16 * r * k]);
        dft8(u);
        wave_lds_fence();
        const int j = (lane - k) * 8 + k;
#pragma unroll
        for (int s = 0; s < 8; ++s) buf[padidx(j + 8 * s)] = u[s];
        wave_lds_fence();
      }
      // pass 3: p = 64
      {
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = buf[padidx(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], s_tw[2 * r * lane]);
        dft8(u);
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < 8; ++s) buf[padidx(lane + 64 * s)] = u[s];
        wave_lds_fence();
      }
      // unpack the real FFT and take the power:  X[k] = E[k] + w^k O[k]
      // (the power spectrum overwrites the front of this wave's FFT buffer: read everything first)
      float pk[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int k = lane + 64 * m;
        const float2 zk = buf[padidx(k)];
        const float2 zn = buf[padidx((kHalf - k) & (kHalf - 1))];
        const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const float2 xk = cadd(e, cmul(s_tw[k], o));
        pk[m] = xk.x * xk.x + xk.y * xk.y;
      }
      wave_lds_fence();
#pragma unroll
      for (int m = 0; m < 8; ++m) pw[lane + 64 * m] = pk[m];
      wave_lds_fence();
      // banded mel filterbank + log + masking + normalisation
      for (int m = lane; m < n_mels; m += 64) {
        const int s0 = s_bs[m];
        float acc = 0.0f;
        for (int j = 0; j < band_len; ++j) acc = fmaf(s_bw[j * n_mels + m], pw[s0 + j], acc);
        float v = logf(acc + 0.00001f);
        if ((m >= mf0 && m < mf1) || (t >= mt0 && t < mt1)) v = 0.0f;
        s_out[m * (kFramesPerBlock + 1) + fl] = (v + 4.5f) / 5.0f;
      }
      wave_lds_fence();
      if (have_next) {   // rotate the prefetched frame in only now, so its loads had the whole FFT to land
#pragma unroll
        for (int i = 0; i < 16; ++i) { cur_lo[i] = nxt_lo[i]; cur_hi[i] = nxt_hi[i]; }
      }
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
                           int n_fft, int hop, const float* twiddle, const float* band_w,
                           const int* band_start, int n_mels, int band_len, float* out, int T,
                           int mask_f0, int mask_f1, int mask_t0, int mask_t1, eat_stream_t stream) {
  eat::clear_stale_error();
  if (n_fft != kNfft) return eat::fail(EAT_EINVAL, "eat_mel_fwd: only n_fft=1024 is implemented (got %d)", n_fft);
  if (win_length < 1 || win_length > n_fft || hop < 1 || B < 1 || n_mels < 1 || band_len < 1)
    return eat::fail(EAT_EINVAL, "eat_mel_fwd: bad geometry");
  if (L - 1 <= n_fft / 2) return eat::fail(EAT_EINVAL, "eat_mel_fwd: clip too short for reflect padding (L=%d)", L);
  if (T != 1 + (L - 1) / hop) return eat::fail(EAT_EINVAL, "eat_mel_fwd: T=%d does not match L=%d hop=%d", T, L, hop);
  size_t smem = sizeof(float2) * (kNfft + kWavesPerBlock * kBufStride) +
                sizeof(float) * (kNfft + (size_t)band_len * n_mels) +
                sizeof(int) * n_mels + sizeof(float) * (size_t)n_mels * (kFramesPerBlock + 1);
  if (smem > 160 * 1024) return eat::fail(EAT_EINVAL, "eat_mel_fwd: mel table too large for LDS (%zu B)", smem);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mel_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eat::fail(EAT_ELAUNCH, "eat_mel_fwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  dim3 grid((T + kFramesPerBlock * kGroupsPerBlock - 1) / (kFramesPerBlock * kGroupsPerBlock), B);
  hipLaunchKernelGGL(mel_fwd_kernel, grid, dim3(256), smem, (hipStream_t)stream, wave, L, window, win_length,
                     hop, reinterpret_cast<const float2*>(twiddle), band_w, band_start, n_mels, band_len, out, T,
                     mask_f0, mask_f1, mask_t0, mask_t1);
  return eat::check_launch("eat_mel_fwd");
}
