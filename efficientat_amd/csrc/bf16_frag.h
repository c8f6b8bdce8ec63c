// Fragment helpers shared by the register-resident bf16 1x1 kernels (conv_pw_stream.hip, expand_dw.hip): the split of
// fp32 activations into bf16 hi / lo B fragments, the MFMA issue order of one 32-row chunk, raw buffer access.
#pragma once
#include "eat_common.h"

namespace eatfrag {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

constexpr int kKC = 32;
constexpr int kTileN = 256;

// 8 rows x 4 consecutive columns of fp32 (rows k0+i of the lane's k-octet) -> the lane's B fragments of 4 n-tiles:
// bh[j] = bf16(x) and bl[j] = bf16(x - bh[j]) of column j (round-to-nearest-even, exactly conv_pw_bf16.hip's split)
template <int NPROD>
__device__ __forceinline__ void split_rows(const float4 (&xr)[8], bf16x8 (&bh)[4], bf16x8 (&bl)[4]) {
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
        bl[j][i] = l[0]; bl[j][i + 1] = l[1];
      }
    }
  }
}

template <int NPROD>
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[4], const bf16x8 ah, const bf16x8 al, const bf16x8 (&bh)[4],
                                           const bf16x8 (&bl)[4]) {
  // the products of one accumulator are issued 4 MFMAs apart (conv_pw_bf16.hip)
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[j], 0, 0, 0);
  if constexpr (NPROD == 3) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[j], 0, 0, 0);
  }
}

// Raw buffer access (irb.hip / dw_plane.hip): per-lane byte offset + scalar offset; a lane whose offset is kOOB stores
// nothing (hardware range check), so the row loop has no divergent branch and the compiler can COUNT the stores it
// leaves in flight (a branch around a store makes it fall back to s_waitcnt vmcnt(0) at the loop head, which drains the
// output stream once per m-tile).
constexpr unsigned kOOB = 0x80000000u;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
  const int n = bytes < 0x7fffffffLL ? (int)bytes : 0x7fffffff;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_load_frag(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store4(float4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 0);
}

}  // namespace eatfrag
