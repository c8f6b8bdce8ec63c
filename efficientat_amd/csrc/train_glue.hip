// Training-loop glue of ex_audioset.py on the device: the reference builds these from ~25 tiny
// torch ops per step and reads three scalars back to the host every step (ex_audioset.py:142-194).
//   eat_mixup_fwd        x[b] * lam[b] + x[perm[b]] * (1 - lam[b])                      (ex_audioset.py:142-148)
//   eat_kd_loss_fwd_bwd  hard-label BCE-with-logits on the mixed targets + knowledge-distillation BCE against the
//                        (mixed) teacher probabilities, lambda-weighted, AND its gradient w.r.t. the logits, in one
//                        pass over the (B, C) logits (ex_audioset.py:149-189)
//   eat_wave_i16_to_f32  16-bit PCM transport of the waveforms: int16 over PCIe, fp32 for the log-mel
#include "eat_common.h"

namespace {

__global__ __launch_bounds__(256) void mixup_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                    const float* __restrict__ lam, float* __restrict__ out, int n4) {
  const int b = blockIdx.y;
  const float l = lam[b], m = 1.0f - l;
  const float4* xa = reinterpret_cast<const float4*>(x) + (size_t)b * n4;
  const float4* xb = reinterpret_cast<const float4*>(x) + (size_t)perm[b] * n4;
  float4* o = reinterpret_cast<float4*>(out) + (size_t)b * n4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const float4 p = xa[i], q = xb[i];
    o[i] = make_float4(p.x * l + q.x * m, p.y * l + q.y * m, p.z * l + q.z * m, p.w * l + q.w * m);
  }
}
__global__ __launch_bounds__(256) void mixup_tail_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                         const float* __restrict__ lam, float* __restrict__ out, int n) {
  const int b = blockIdx.y;
  const float l = lam[b], m = 1.0f - l;
  const size_t pa = (size_t)b * n, pb = (size_t)perm[b] * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[pa + i] = x[pa + i] * l + x[pb + i] * m;
}

// 8 samples per lane and trip: one 16-byte load, two 16-byte stores
__global__ __launch_bounds__(256) void wave_i16_kernel(const short* __restrict__ src, float* __restrict__ dst, long long n8,
                                                       float scale) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int4 v = reinterpret_cast<const int4*>(src)[i];
    const int w[4] = {v.x, v.y, v.z, v.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[2 * j] = (float)(short)(w[j] & 0xffff) * scale;
      o[2 * j + 1] = (float)(w[j] >> 16) * scale;
    }
    reinterpret_cast<float4*>(dst)[2 * i] = make_float4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(o[4], o[5], o[6], o[7]);
  }
}
__global__ __launch_bounds__(256) void wave_i16_tail_kernel(const short* __restrict__ src, float* __restrict__ dst, long long n0,
                                                            long long n, float scale) {
  const long long i = n0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i] * scale;
}

// numerically stable BCE-with-logits term (torch: max(z,0) - z t + log1p(exp(-|z|)))
__device__ __forceinline__ float bce(float z, float t) { return fmaxf(z, 0.0f) - z * t + log1pf(expf(-fabsf(z))); }

// one block per sample; sums[0..2] += (total, label part, distillation part) of the batch-mean loss
__global__ __launch_bounds__(256) void kd_loss_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                      const int* __restrict__ perm, const float* __restrict__ lam,
                                                      const float* __restrict__ teacher, const long long* __restrict__ tidx,
                                                      int n_teacher, float kd_lambda, int B, int C,
                                                      float* __restrict__ sums, float* __restrict__ dz) {
  __shared__ float s_red[2][4];
  const int b = blockIdx.x;
  const bool mix = perm != nullptr;
  const int pb = mix ? perm[b] : b;
  const float l = mix ? lam[b] : 1.0f, m = 1.0f - l;
  const bool kd = teacher != nullptr && kd_lambda < 1.0f;
  // the reference indexes teacher_preds[-1] for files without a teacher entry and zeroes THIS sample's KD term; the
  // mix-up partner's row is used whatever its own status is (ex_audioset.py:160-180): reproduced
  long long ia = 0, ib = 0;
  bool known = false;
  if (kd) {
    ia = tidx[b]; ib = tidx[pb];
    known = ia >= 0;
    ia = ia < 0 ? n_teacher - 1 : ia;
    ib = ib < 0 ? n_teacher - 1 : ib;
  }
  const float wl = kd ? kd_lambda : 1.0f, wk = kd ? (1.0f - kd_lambda) * (known ? 1.0f : 0.0f) : 0.0f;
  const float inv = 1.0f / ((float)B * (float)C);
  float s_label = 0.0f, s_kd = 0.0f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float zz = z[(size_t)b * C + c];
    const float ym = y[(size_t)b * C + c] * l + (mix ? y[(size_t)pb * C + c] * m : 0.0f);
    const float sg = 1.0f / (1.0f + expf(-zz));
    s_label += bce(zz, ym);
    float g = wl * (sg - ym);
    if (kd) {
      const float ta = teacher[(size_t)ia * C + c], tb = teacher[(size_t)ib * C + c];
      s_kd += bce(zz, ta) * l + bce(zz, tb) * m;
      g += wk * (sg - (ta * l + tb * m));
    }
    dz[(size_t)b * C + c] = g * inv;
  }
  s_label = eat::wave_sum(s_label);
  s_kd = eat::wave_sum(s_kd);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wv] = s_label; s_red[1][wv] = s_kd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = (s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]) * inv * wl;
    const float k = (s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]) * inv * wk;
    atomicAdd(sums + 0, a + k);
    atomicAdd(sums + 1, a);
    atomicAdd(sums + 2, k);
  }
}

// ---- out[c] = sum_r m[r, c] for a row-major (R, C) matrix: the bias gradients of the context-path Linear layers of DyMN
// (models/dymn/dy_block.py:235-254; R = B * (F + T) up to 64 k rows).  Consecutive threads read consecutive elements
// (fully coalesced whatever C is), every element goes into a per-block LDS accumulator of its column, one global atomic
// per column and block.  (The 1 x R times R x C product on the linear kernel that this replaces ran on ONE block.)
// Many rows (DyMN context path: R = B (F + T) ~ 72 k rows of a few hundred columns): a thread owns one column of its
// block's row range and adds in registers (4 row groups x 64 columns per block, coalesced 256-byte row segments), one LDS
// combination and one atomic per column and block.  (First version: one LDS atomic per ELEMENT - 35 us per call, 91 calls per
// dymn20 step.)
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ m, float* __restrict__ out, int R, int C,
                                                      int rows_per_block) {
  __shared__ float s_red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = (r0 + rows_per_block) < R ? (r0 + rows_per_block) : R;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (c < C) {
    int r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
      a0 += m[(size_t)r * C + c]; a1 += m[(size_t)(r + 4) * C + c];
      a2 += m[(size_t)(r + 8) * C + c]; a3 += m[(size_t)(r + 12) * C + c];
    }
    for (; r < r1; r += 4) a0 += m[(size_t)r * C + c];
  }
  s_red[ty][tx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ty == 0 && c < C) atomicAdd(out + c, (s_red[0][tx] + s_red[1][tx]) + (s_red[2][tx] + s_red[3][tx]));
}

// Few rows (R <= 512: the per-(b,c) plane sums of the train plan, R = batch): no atomics, bit-reproducible (these column sums
// feed BatchNorm statistics: a last-bit difference decides on which side of an activation kink some element falls, so
// run-to-run noise here becomes 1e-3-level gradient noise).  A block = 32 columns x 8 row groups: thread (g, c) adds rows
// g, g + 8, ... in index order, the 8 partial sums are added in a fixed tree.  (Round 3: one thread per column walking all
// 256 rows - 18.9 us per launch in the captured step, 14 launches.)
__global__ __launch_bounds__(256) void col_sum_det_kernel(const float* __restrict__ m, float* __restrict__ out, int R, int C) {
  __shared__ float s_part[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a0 = 0.0f, a1 = 0.0f;
  if (c < C) {
    int r = g;
    for (; r + 8 < R; r += 16) { a0 += m[(size_t)r * C + c]; a1 += m[(size_t)(r + 8) * C + c]; }
    if (r < R) a0 += m[(size_t)r * C + c];
  }
  s_part[g][cl] = a0 + a1;
  __syncthreads();
  if (g == 0 && c < C)
    out[c] = ((s_part[0][cl] + s_part[1][cl]) + (s_part[2][cl] + s_part[3][cl])) +
             ((s_part[4][cl] + s_part[5][cl]) + (s_part[6][cl] + s_part[7][cl]));
}

// ---- calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (measurement support, not on the hot path): copies of
// a KNOWN byte count with the access widths the library's kernels use.  On gfx950 FETCH_SIZE reports half the bytes of wide
// coalesced reads (MI355X_MICROARCH.md, HBM section); bench.py divides the known bytes of these launches by the counter
// reading taken in the SAME rocprofv3 pass as the kernels it corrects.
//   MODE 0: 16 B per lane global loads     1: 16 B per lane LDS-DMA loads (global_load_lds_dwordx4)
//        2: 4 B per lane loads              3: 8 B per lane loads           (stores: same width as the loads; LDS-DMA: 16 B)
typedef __attribute__((address_space(3))) void calib_lds_void;
template <int MODE>
__global__ __launch_bounds__(256) void calib_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if constexpr (MODE == 0) {
    for (long long i = t; i * 4 + 3 < n; i += stride)
      reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  } else if constexpr (MODE == 1) {
    __shared__ __attribute__((aligned(16))) float s_buf[4][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long long i = t; i * 4 + 3 < n; i += stride) {
      const unsigned dstl = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(calib_lds_void*)&s_buf[wv][0]);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_waitcnt vmcnt(0)"
                   ::"s"(dstl), "v"(reinterpret_cast<const float4*>(src) + i) : "memory", "m0");
      reinterpret_cast<float4*>(dst)[i] = *reinterpret_cast<const float4*>(&s_buf[wv][lane * 4]);
    }
  } else if constexpr (MODE == 2) {
    for (long long i = t; i < n; i += stride) dst[i] = src[i];
  } else {
    for (long long i = t; i * 2 + 1 < n; i += stride)
      reinterpret_cast<float2*>(dst)[i] = reinterpret_cast<const float2*>(src)[i];
  }
}

}  // namespace

// n floats (a multiple of 4) from src to dst with the access width of `mode` (see calib_copy_kernel): 4 n bytes read, 4 n written
extern "C" int eat_calib_copy(const float* src, float* dst, long long n, int mode, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!src || !dst || n < 4 || (n & 3) || mode < 0 || mode > 3) return eat::fail(EAT_EINVAL, "eat_calib_copy: bad arguments");
  const dim3 grid(256 * 16), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(calib_copy_kernel<0>, grid, blk, 0, s, src, dst, n);
  else if (mode == 1) hipLaunchKernelGGL(calib_copy_kernel<1>, grid, blk, 0, s, src, dst, n);
  else if (mode == 2) hipLaunchKernelGGL(calib_copy_kernel<2>, grid, blk, 0, s, src, dst, n);
  else hipLaunchKernelGGL(calib_copy_kernel<3>, grid, blk, 0, s, src, dst, n);
  return eat::check_launch("eat_calib_copy");
}

extern "C" int eat_mixup_fwd(const float* x, const int* perm, const float* lam, float* out, int B, int n,
                             eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || n < 1) return eat::fail(EAT_EINVAL, "eat_mixup_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if ((n & 3) == 0) {
    const int n4 = n >> 2;
    int gx = (n4 + 255) / 256;
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(mixup_kernel, dim3(gx, B), dim3(256), 0, s, x, perm, lam, out, n4);
  } else {
    int gx = (n + 255) / 256;
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(mixup_tail_kernel, dim3(gx, B), dim3(256), 0, s, x, perm, lam, out, n);
  }
  return eat::check_launch("eat_mixup_fwd");
}

extern "C" int eat_wave_i16_to_f32(const short* src, float* dst, long long n, float scale, eat_stream_t stream) {
  eat::clear_stale_error();
  if (!src || !dst || n < 1) return eat::fail(EAT_EINVAL, "eat_wave_i16_to_f32: bad arguments");
  if (((size_t)src & 15) || ((size_t)dst & 15)) return eat::fail(EAT_EINVAL, "eat_wave_i16_to_f32: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long long n8 = n >> 3;
  if (n8 > 0) {
    long long blocks = (n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(wave_i16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n8, scale);
  }
  if (n & 7) hipLaunchKernelGGL(wave_i16_tail_kernel, dim3(1), dim3(256), 0, s, src, dst, n8 << 3, n, scale);
  return eat::check_launch("eat_wave_i16_to_f32");
}

extern "C" int eat_kd_loss_fwd_bwd(const float* logits, const float* y, const int* perm, const float* lam,
                                   const float* teacher, const long long* teacher_idx, int n_teacher, float kd_lambda,
                                   int B, int C, float* sums, float* dlogits, eat_stream_t stream) {
  eat::clear_stale_error();
  if (B < 1 || C < 1) return eat::fail(EAT_EINVAL, "eat_kd_loss_fwd_bwd: bad shape");
  if ((perm == nullptr) != (lam == nullptr)) return eat::fail(EAT_EINVAL, "eat_kd_loss_fwd_bwd: perm and lam go together");
  if (teacher && (!teacher_idx || n_teacher < 1)) return eat::fail(EAT_EINVAL, "eat_kd_loss_fwd_bwd: teacher needs its index");
  if (kd_lambda < 0.0f || kd_lambda > 1.0f) return eat::fail(EAT_EINVAL, "eat_kd_loss_fwd_bwd: kd_lambda outside [0, 1]");
  hipLaunchKernelGGL(kd_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, y, perm, lam, teacher, teacher_idx,
                     n_teacher, kd_lambda, B, C, sums, dlogits);
  return eat::check_launch("eat_kd_loss_fwd_bwd");
}

namespace {
__global__ __launch_bounds__(256) void col_zero_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.0f;
}
}  // namespace

extern "C" int eat_col_sum(const float* m, float* out, int R, int C, eat_stream_t stream) {
  eat::clear_stale_error();
  if (R < 1 || C < 1 || C > 8192) return eat::fail(EAT_EINVAL, "eat_col_sum: bad shape (%d x %d)", R, C);
  hipStream_t s = (hipStream_t)stream;
  if (R <= 512) {                  // the train plan's per-(b,c) plane sums (R = batch): fixed order, no atomics
    hipLaunchKernelGGL(col_sum_det_kernel, dim3((C + 31) / 32), dim3(256), 0, s, m, out, R, C);
    return eat::check_launch("eat_col_sum");
  }
  // (a kernel, not hipMemsetAsync: memset nodes of a captured hipGraph were not ordered reliably - see eat_dyn_bank_grad)
  hipLaunchKernelGGL(col_zero_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, out, C);
  const int cb = (C + 63) / 64;
  int rows_pb = (int)(((long long)R * cb + 1023) / 1024);             // ~1024 blocks, at least 64 rows each
  if (rows_pb < 64) rows_pb = 64;
  hipLaunchKernelGGL(col_sum_kernel, dim3((unsigned)cb, (unsigned)((R + rows_pb - 1) / rows_pb)), dim3(256), 0, s, m, out, R, C,
                     rows_pb);
  return eat::check_launch("eat_col_sum");
}

// ---- multi-tensor Adam / AdamW: the optimizer step of ex_audioset.py:86-91,197-199 (torch.optim.Adam / AdamW over all
// parameters) as ONE launch over a device-resident chunk table - every block owns one chunk (<= 4096
// consecutive elements of one parameter), so small BatchNorm vectors and multi-MB weight matrices share a launch without idle
// blocks.  The step counter and, optionally, the learning rate live on the device (hipGraph replays / LR schedulers that write a
// tensor): the kernel reads them, a one-thread kernel advances the counter afterwards.
//   g' = g * grad_scale (+ wd * p, Adam's L2 form);  AdamW: p *= 1 - lr * wd
//   m = m + (1 - b1) (g' - m);  v = b2 v + (1 - b2) g'^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// - the order of operations of torch's fused kernel (aten/src/ATen/native/cuda/fused_adam_utils.cuh), bias corrections in fp64.
namespace {
struct AdamChunk { float* p; const float* g; float* m; float* v; int n; int pad; };
static_assert(sizeof(AdamChunk) == 40, "host side builds the table as 40-byte records");

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamChunk* __restrict__ table, const float* __restrict__ lr_ptr,
                                                         double lr_val, const float* __restrict__ step_ptr, float step_val,
                                                         double b1, double b2, double eps, double wd, int decoupled,
                                                         double grad_scale) {
  const AdamChunk c = table[blockIdx.x];
  const double t = (double)(step_ptr ? *step_ptr : step_val) + 1.0;
  const double lr = lr_ptr ? (double)*lr_ptr : lr_val;
  const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
  const double step_size = lr / bc1, bc2_sqrt = sqrt(bc2);
  const float w1 = (float)(1.0 - b1), gsc = (float)grad_scale;
  const double decay = 1.0 - lr * wd, omb2 = 1.0 - b2;
  // the expression types of torch's fused kernel: the first moment is a float lerp, every line that multiplies by a double
  // hyper-parameter (second moment, denominator, update) is evaluated in fp64 and rounded on assignment
  auto upd = [&](float& p, float g, float& m, float& v) {
    g *= gsc;
    if (decoupled) p = (float)((double)p * decay); else if (wd != 0.0) g = (float)((double)g + wd * (double)p);
    m = fmaf(w1, g - m, m);
    v = (float)(b2 * (double)v + omb2 * (double)g * (double)g);
    const double denom = (double)sqrtf(v) / bc2_sqrt + eps;
    p = (float)((double)p - step_size * (double)m / denom);
  };
  const bool vec = ((c.n & 3) == 0) && (((size_t)c.p | (size_t)c.g | (size_t)c.m | (size_t)c.v) & 15) == 0;
  if (vec) {
    for (int i = threadIdx.x * 4; i < c.n; i += 1024) {
      float4 p = *reinterpret_cast<float4*>(c.p + i), m = *reinterpret_cast<float4*>(c.m + i), v = *reinterpret_cast<float4*>(c.v + i);
      const float4 g = *reinterpret_cast<const float4*>(c.g + i);
      upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
      *reinterpret_cast<float4*>(c.p + i) = p; *reinterpret_cast<float4*>(c.m + i) = m; *reinterpret_cast<float4*>(c.v + i) = v;
    }
  } else {
    for (int i = threadIdx.x; i < c.n; i += 256) {
      float p = c.p[i], m = c.m[i], v = c.v[i];
      upd(p, c.g[i], m, v);
      c.p[i] = p; c.m[i] = m; c.v[i] = v;
    }
  }
}

__global__ void adam_step_inc_kernel(float* __restrict__ step) { *step += 1.0f; }
}  // namespace

extern "C" int eat_adam_multi(const void* table, int n_chunks, const float* lr_ptr, double lr, float* step_ptr, float step,
                              double beta1, double beta2, double eps, double weight_decay, int decoupled, double grad_scale,
                              eat_stream_t stream) {
  eat::clear_stale_error();
  if (!table || n_chunks < 1) return eat::fail(EAT_EINVAL, "eat_adam_multi: empty chunk table");
  if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
    return eat::fail(EAT_EINVAL, "eat_adam_multi: bad hyper-parameters (beta1=%g beta2=%g eps=%g)", beta1, beta2, eps);
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const AdamChunk*>(table), lr_ptr, lr, (const float*)step_ptr, step, beta1, beta2, eps,
                     weight_decay, decoupled, grad_scale);
  if (step_ptr) hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_ptr);
  return eat::check_launch("eat_adam_multi");
}
