// Skinny GEMM for incremental decoding (caption generation with a K/V cache, reference models/model.py:657-675 -> HF generate
// with use_cache): y[M, N] = epilogue(x[M, K] . W[K, N]) for a HANDFUL of rows (M = images x beams, 1..64).
//
// The 128^2 / 256^2 tile kernels give such a product N/128 workgroups (8-32 of 256 CUs), each walking the whole K behind
// one another's load latency: 30 us per GEMM, 3.2 ms per token for GPT-2-M whatever the batch (tools/decode_probe.py).  The
// work is one pass over the WEIGHTS (25 MB per layer), so it is laid out for the memory system instead:
//   pass 1 (skinny_nn_kernel): grid = column chunks x K splits x row chunks of ONE-wave workgroups (200-1000 of them); a
//          wave streams whole 1-KiB rows of W (16 B per lane, 8 rows in flight), multiplies them into <= 16 rows of x held
//          in LDS as f32 (VALU: 16 FMAs per weight element keeps up with HBM) and STORES its partial [rows, 512] block of
//          its K split's [M, N] slab (plain coalesced stores: neither LDS float atomics between waves - ~1 lane per clock
//          - nor global float atomics between the K splits - 32-128 workgroups on the same addresses - were fast);
//   pass 2 (skinny_finish_kernel): one workgroup per row sums the K splits' slabs (fixed order: reproducible), applies bias
//          / gelu_new / f32 residual, writes the f32 and/or bf16 result (arbitrary row strides: the K/V cache row, the
//          residual stream) and optionally runs the NEXT LayerNorm on the finished row (its bf16 output is the next
//          product's operand: no separate LN launch).
#include "common.h"

using namespace pgca;

namespace {

constexpr int SK_MT = 16;       // rows of x per pass
constexpr int SK_COLS = 512;    // columns per workgroup: 64 lanes x 8 bf16 (one 16-B load per lane per row of W)
constexpr int SK_MAXKC = 256;   // rows of W per workgroup at most (x slice in LDS: 256 x 16 f32 = 16 KiB)

// NQ = groups of 4 rows per pass (1..4): a template parameter so that the accumulators are statically indexed registers.
// ONE wave per workgroup: its 64 lanes own 512 columns of `kchunk` rows of W outright, so nothing is reduced across
// waves (LDS float atomics for that were measured at ~1 lane per clock: 4 us per product per 4 rows).
template <int NQ>
__global__ __launch_bounds__(64) void skinny_nn_kernel(const bf16_t* __restrict__ x, int lda,
                                                       const bf16_t* __restrict__ W, int ldw, int M, int N, int K,
                                                       int kchunk, float* __restrict__ scratch) {
  __shared__ __attribute__((aligned(16))) float xs[SK_MAXKC * SK_MT];   // [k][m], m fastest
  const int lane = threadIdx.x;
  const int c0 = blockIdx.x * SK_COLS + lane * 8;
  const int k0 = blockIdx.y * kchunk;
  const int kn = min(kchunk, K - k0);
  const int m0 = blockIdx.z * SK_MT;
  const int mt = min(SK_MT, M - m0);
  for (int m = 0; m < 4 * NQ; ++m)               // coalesced along k, one row of x after the other
    for (int k = lane; k < kn; k += 64)
      xs[k * SK_MT + m] = m < mt ? bf2f(x[(size_t)(m0 + m) * lda + k0 + k]) : 0.f;
  __syncthreads();
  float acc[4 * NQ][8];
#pragma unroll
  for (int m = 0; m < 4 * NQ; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  const bool colok = c0 < N;   // N % 8 == 0: a lane's 8 columns are all inside or all outside
  const bf16_t* wp = W + (size_t)k0 * ldw + c0;
  constexpr int U = 8;          // rows of W in flight
  for (int kb = 0; kb < kn; kb += U) {
    u32x4 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      w[u] = (u32x4){0u, 0u, 0u, 0u};
      if (colok && kb + u < kn) w[u] = *reinterpret_cast<const u32x4*>(wp + (size_t)(kb + u) * ldw);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kb + u < kn) {   // wave-uniform
        const bf16x8 wv = __builtin_bit_cast(bf16x8, w[u]);
        float wf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = bf2f(wv[j]);
        const float4* xp = reinterpret_cast<const float4*>(xs + (kb + u) * SK_MT);   // same address in every lane: broadcast
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float4 xv = xp[q];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[4 * q + 0][j] += xv.x * wf[j];
            acc[4 * q + 1][j] += xv.y * wf[j];
            acc[4 * q + 2][j] += xv.z * wf[j];
            acc[4 * q + 3][j] += xv.w * wf[j];
          }
        }
      }
    }
  }
  if (!colok) return;
  float* slab = scratch + (size_t)blockIdx.y * M * N + c0;   // this K split's partial [M, N]: 1 KiB per row per wave
#pragma unroll
  for (int m = 0; m < 4 * NQ; ++m) {
    if (m < mt) {
      float4* p = reinterpret_cast<float4*>(slab + (size_t)(m0 + m) * N);
      p[0] = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
      p[1] = make_float4(acc[m][4], acc[m][5], acc[m][6], acc[m][7]);
    }
  }
}

// One workgroup per row.  NVF = float4 per thread (N <= 256 * 4 * NVF).
template <int NVF>
__global__ __launch_bounds__(256) void skinny_finish_kernel(const float* __restrict__ scratch, int ksplit, int M, int N,
                                                            const float* __restrict__ bias,
                                                            int act, const float* __restrict__ residual, int ld_res,
                                                            float* __restrict__ out_f32, int ld_out_f32,
                                                            bf16_t* __restrict__ out_bf16, int ld_out_bf16,
                                                            const float* __restrict__ ln_gamma,
                                                            const float* __restrict__ ln_beta, float ln_eps,
                                                            bf16_t* __restrict__ ln_out, int ld_ln) {
  __shared__ float redw[8];
  const int m = blockIdx.x, t = threadIdx.x;
  float4 v[NVF];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NVF; ++i) {
    const int c = (i * 256 + t) * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < N) {
      const float* sp = scratch + (size_t)m * N + c;
#pragma unroll 4
      for (int ks = 0; ks < ksplit; ++ks) {
        const float4 p = *reinterpret_cast<const float4*>(sp + (size_t)ks * M * N);
        v[i].x += p.x; v[i].y += p.y; v[i].z += p.z; v[i].w += p.w;
      }
      if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + c);
        v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w;
      }
      if (act == PGCA_EPI_GELU_NEW) {
        v[i].x = gelu_new(v[i].x); v[i].y = gelu_new(v[i].y); v[i].z = gelu_new(v[i].z); v[i].w = gelu_new(v[i].w);
      }
      if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + (size_t)m * ld_res + c);
        v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
      }
      if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)m * ld_out_f32 + c) = v[i];
      if (out_bf16) {
        bf16x4 o;
        o[0] = f2bf(v[i].x); o[1] = f2bf(v[i].y); o[2] = f2bf(v[i].z); o[3] = f2bf(v[i].w);
        *reinterpret_cast<bf16x4*>(out_bf16 + (size_t)m * ld_out_bf16 + c) = o;
      }
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  if (!ln_out) return;   // uniform
  // LayerNorm of the finished row (two-pass statistics like F.layer_norm / pgca_layernorm_fwd)
  s = wave_sum(s);
  if ((t & 63) == 0) redw[t >> 6] = s;
  __syncthreads();
  const float mean = (redw[0] + redw[1] + redw[2] + redw[3]) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NVF; ++i) {
    const int c = (i * 256 + t) * 4;
    if (c < N) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  q = wave_sum(q);
  if ((t & 63) == 0) redw[4 + (t >> 6)] = q;
  __syncthreads();
  const float rstd = rsqrtf((redw[4] + redw[5] + redw[6] + redw[7]) / (float)N + ln_eps);
#pragma unroll
  for (int i = 0; i < NVF; ++i) {
    const int c = (i * 256 + t) * 4;
    if (c < N) {
      const float4 g = *reinterpret_cast<const float4*>(ln_gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(ln_beta + c);
      bf16x4 o;
      o[0] = f2bf((v[i].x - mean) * rstd * g.x + b.x);
      o[1] = f2bf((v[i].y - mean) * rstd * g.y + b.y);
      o[2] = f2bf((v[i].z - mean) * rstd * g.z + b.z);
      o[3] = f2bf((v[i].w - mean) * rstd * g.w + b.w);
      *reinterpret_cast<bf16x4*>(ln_out + (size_t)m * ld_ln + c) = o;
    }
  }
}

}  // namespace

// K split plan: enough workgroups to put every CU's memory pipeline to work, but >= 32 rows of W each (the x slice and the
// LDS reduction are per workgroup)
static void skinny_plan(int M, int N, int K, int* ncol, int* nz, int* ksplit, int* kchunk) {
  *ncol = (N + SK_COLS - 1) / SK_COLS;
  *nz = (M + SK_MT - 1) / SK_MT;
  int ks = 1024 / (*ncol * *nz);
  ks = ks < 1 ? 1 : ks;
  int kc = (K + ks - 1) / ks;
  kc = kc < 32 ? 32 : kc;
  kc = (kc + 15) / 16 * 16;
  kc = kc > SK_MAXKC ? SK_MAXKC : kc;
  *kchunk = kc;
  *ksplit = (K + kc - 1) / kc;
}

extern "C" int64_t pgca_gemm_skinny_workspace(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int ncol, nz, ksplit, kchunk;
  skinny_plan(M, N, K, &ncol, &nz, &ksplit, &kchunk);
  return (int64_t)ksplit * M * N * (int64_t)sizeof(float);
}

extern "C" int pgca_gemm_skinny(const pgca_skinny_args* args, void* stream) {
  if (!args) {
    set_error("pgca_gemm_skinny: null arguments");
    return PGCA_ERR_INVALID;
  }
  const pgca_skinny_args& a = *args;
  const bool al = !(((uintptr_t)a.x | (uintptr_t)a.W | (uintptr_t)a.scratch | (uintptr_t)a.bias | (uintptr_t)a.residual |
                     (uintptr_t)a.out_f32 | (uintptr_t)a.out_bf16 | (uintptr_t)a.ln_out_bf16 | (uintptr_t)a.ln_gamma |
                     (uintptr_t)a.ln_beta) & 15);
  if (!a.x || !a.W || !a.scratch || a.M <= 0 || a.M > PGCA_SKINNY_MAX_M || a.N <= 0 || a.K <= 0 || (a.N & 7) ||
      (a.ldw & 7) || a.ldw < a.N || a.lda < a.K || !al || (!a.out_f32 && !a.out_bf16) ||
      (a.act != PGCA_EPI_NONE && a.act != PGCA_EPI_GELU_NEW) || a.N > 8192 || (a.residual && (a.ld_res & 3)) ||
      (a.out_f32 && (a.ld_out_f32 & 3)) || (a.out_bf16 && (a.ld_out_bf16 & 3)) ||
      (a.ln_out_bf16 && (!a.ln_gamma || !a.ln_beta || (a.ld_ln & 3)))) {
    set_error("pgca_gemm_skinny: bad arguments (M=%d <= %d, N=%d %% 8 == 0 and <= 8192, K=%d, 16-B aligned pointers, "
              "row strides multiples of 4)", a.M, PGCA_SKINNY_MAX_M, a.N, a.K);
    return PGCA_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  int ncol, nz, ksplit, kchunk;
  skinny_plan(a.M, a.N, a.K, &ncol, &nz, &ksplit, &kchunk);
  const int nq = ((a.M < SK_MT ? a.M : SK_MT) + 3) / 4;   // row groups of 4 a pass really holds
#define SK_PASS1(NQV)                                                                                                  \
  hipLaunchKernelGGL((skinny_nn_kernel<NQV>), dim3(ncol, ksplit, nz), dim3(64), 0, s, (const bf16_t*)a.x, a.lda,       \
                     (const bf16_t*)a.W, a.ldw, a.M, a.N, a.K, kchunk, a.scratch)
  switch (nq) {
    case 1: SK_PASS1(1); break;
    case 2: SK_PASS1(2); break;
    case 3: SK_PASS1(3); break;
    default: SK_PASS1(4); break;
  }
#undef SK_PASS1
  const int nvf = (a.N + 1023) / 1024;
#define SK_FINISH(NV)                                                                                                      \
  hipLaunchKernelGGL((skinny_finish_kernel<NV>), dim3(a.M), dim3(256), 0, s, a.scratch, ksplit, a.M, a.N, a.bias, a.act, a.residual, \
                     a.ld_res, a.out_f32, a.ld_out_f32, (bf16_t*)a.out_bf16, a.ld_out_bf16, a.ln_gamma, a.ln_beta, a.ln_eps, \
                     (bf16_t*)a.ln_out_bf16, a.ld_ln)
  switch (nvf) {
    case 1: SK_FINISH(1); break;
    case 2: SK_FINISH(2); break;
    case 3: SK_FINISH(3); break;
    case 4: SK_FINISH(4); break;
    case 5: SK_FINISH(5); break;
    case 6: SK_FINISH(6); break;
    case 7: SK_FINISH(7); break;
    default: SK_FINISH(8); break;
  }
#undef SK_FINISH
  return check_launch("pgca_gemm_skinny");
}
