// Storage type of an activation tensor in HBM: fp32, or bf16 for the WIDE tensors of the mixed-precision training plan
// (BASELINE configs[2]; the reference's `precision=16` / autocast surface, ex_pl_audioset.py:287-293: conv outputs and their
// gradients live in 16-bit storage, statistics / parameters / optimizer state in fp32).  Arithmetic is fp32 in registers
// whatever the storage; a bf16 store rounds to nearest even (v_cvt_pk_bf16_f32), a bf16 load is exact.
#pragma once
#include <hip/hip_runtime.h>

namespace eat {

using bf16_t = __bf16;
using io_u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using io_f32x2 = __attribute__((ext_vector_type(2))) float;
using io_bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {          // (a -> low half, b -> high half), round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector(io_f32x2{a, b}, io_bf16x2));
}
// what a value becomes once it has been stored as bf16 and read back (statistics of a stored tensor are taken of THIS)
__device__ __forceinline__ float bf_round(float v) { return (float)(bf16_t)v; }

template <typename T> struct Io;
template <> struct Io<float> {
  static constexpr int kB = 4;
  static constexpr bool kBf = false;
  static __device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ float load1(const float* p) { return *p; }
  static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Io<bf16_t> {
  static constexpr int kB = 2;
  static constexpr bool kBf = true;
  static __device__ __forceinline__ float4 load4(const bf16_t* p) {       // 8-byte aligned (element index % 4 == 0)
    const io_u32x2 w = *reinterpret_cast<const io_u32x2*>(p);
    return make_float4(bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1]));
  }
  static __device__ __forceinline__ void store4(bf16_t* p, float4 v) {
    *reinterpret_cast<io_u32x2*>(p) = io_u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
  }
  static __device__ __forceinline__ float load1(const bf16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void store1(bf16_t* p, float v) { *p = (bf16_t)v; }
  static __device__ __forceinline__ float rnd(float v) { return bf_round(v); }
};

}  // namespace eat
