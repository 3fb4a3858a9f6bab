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

// sigma(z) with one v_exp and one v_rcp (saturates correctly: exp -> inf => 0, exp -> 0 => 1).
__device__ __forceinline__ float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float fast_tanh(float u) { return 2.0f * fast_sigmoid(2.0f * u) - 1.0f; }
// HF NewGELUActivation (transformers/activations.py:59-66): 0.5 x (1 + tanh(u)) == x sigma(2u),
// u = sqrt(2/pi) (x + 0.044715 x^3).  Written through sigma so the epilogue costs ~8 VALU ops per element.
__device__ __forceinline__ float gelu_new(float x) {
  const float k2 = 2.0f * 0.7978845608028654f;
  return x * fast_sigmoid(k2 * x * (1.0f + 0.044715f * x * x));
}
__device__ __forceinline__ float dgelu_new(float x) {
  const float k2 = 2.0f * 0.7978845608028654f;
  const float x2 = x * x;
  const float s = fast_sigmoid(k2 * x * (1.0f + 0.044715f * x2));
  return s + x * s * (1.0f - s) * k2 * (1.0f + 3.0f * 0.044715f * x2);
}
// both from one sigmoid (forward epilogue PGCA_EPI_GELU_NEW_D)
__device__ __forceinline__ void gelu_new_both(float x, float& y, float& dy) {
  const float k2 = 2.0f * 0.7978845608028654f;
  const float x2 = x * x;
  const float s = fast_sigmoid(k2 * x * (1.0f + 0.044715f * x2));
  y = x * s;
  dy = s + y * (1.0f - s) * k2 * (1.0f + 3.0f * 0.044715f * x2);
}
// HF QuickGELUActivation (transformers/activations.py:117-123)
__device__ __forceinline__ float quick_gelu(float x) { return x * fast_sigmoid(1.702f * x); }
__device__ __forceinline__ float dquick_gelu(float x) {
  const float s = fast_sigmoid(1.702f * x);
  return s + 1.702f * x * s * (1.0f - s);
}

// Counter-based dropout: keep(seed, idx) is a pure function of the site seed and the element's linear index, so the
// backward replays the forward's mask without storing it.  One lowbias32 hash serves a PAIR of elements (idx >> 1):
// the even element takes the low 16 bits, the odd one the high 16; an element is dropped when its 16 bits are below
// threshold >> 16 (threshold = p * 2^32, so p is honoured to 2^-16).  Halving the hashes matters: each costs three
// quarter-rate 32-bit multiplies, and the GEMM epilogues, the attention kernels and the LayerNorm backward replay one
// decision per element.  The oracle restates the same function (oracle/restatement.py: dropout_multiplier).
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
struct Drop {
  unsigned seed, threshold;  // threshold == 0 -> dropout disabled
  float scale;               // 1 / (1 - p)
  __device__ __forceinline__ bool on() const { return threshold != 0u; }
  __device__ __forceinline__ unsigned pair_bits(unsigned pair) const { return hash32(pair * 0x9E3779B1U + seed); }
  __device__ __forceinline__ float mul(unsigned idx) const {  // multiplier of element idx: 0 or 1/(1-p)
    const unsigned h = pair_bits(idx >> 1);
    return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= (threshold >> 16) ? scale : 0.f;
  }
  // elements even_idx and even_idx + 1 from ONE hash (even_idx must be even)
  __device__ __forceinline__ void mul2(unsigned even_idx, float& m0, float& m1) const {
    const unsigned h = pair_bits(even_idx >> 1), t = threshold >> 16;
    m0 = (h & 0xffffu) >= t ? scale : 0.f;
    m1 = (h >> 16) >= t ? scale : 0.f;
  }
  // four consecutive elements: two hashes when base is even (every hot call site), three otherwise
  __device__ __forceinline__ void mul4(unsigned base, float (&m)[4]) const {
    if (!(base & 1u)) {
      mul2(base, m[0], m[1]);
      mul2(base + 2u, m[2], m[3]);
    } else {
      m[0] = mul(base);
      mul2(base + 1u, m[1], m[2]);
      m[3] = mul(base + 3u);
    }
  }
};

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
