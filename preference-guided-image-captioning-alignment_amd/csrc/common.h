// Shared device/host helpers for the gfx950 kernels (wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pgca_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace pgca {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

__device__ __forceinline__ unsigned int pack2(float a, float b) {
  bf16x2 t;
  t[0] = (bf16_t)a;
  t[1] = (bf16_t)b;
  return __builtin_bit_cast(unsigned int, t);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// tanh via exp: 1 - 2/(e^{2u}+1); saturates correctly for |u| large.
__device__ __forceinline__ float fast_tanh(float u) {
  float e = __expf(2.0f * u);
  return 1.0f - 2.0f / (e + 1.0f);
}
// HF NewGELUActivation (transformers/activations.py:59-66)
__device__ __forceinline__ float gelu_new(float x) {
  const float k = 0.7978845608028654f;
  float u = k * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + fast_tanh(u));
}
__device__ __forceinline__ float dgelu_new(float x) {
  const float k = 0.7978845608028654f;
  float x2 = x * x;
  float u = k * (x + 0.044715f * x * x2);
  float t = fast_tanh(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * k * (1.0f + 3.0f * 0.044715f * x2);
}
// HF QuickGELUActivation (transformers/activations.py:117-123)
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

// Two transposed LDS reads -> one MFMA 16x16x32 fragment from a [k][m] (k-strided) image.
// addr0 points at row (kbase + q), addr1 at row (kbase + 4 + q) of the 4x16 blocks (see gemm.hip).
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* addr0, const unsigned char* addr1) {
  i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(addr0));
  i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(addr1));
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  i16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, r);
}

}  // namespace pgca
