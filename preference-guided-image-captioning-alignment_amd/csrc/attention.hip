// Fused multi-head self-attention for head_dim 64 and S <= 128 (GPT-2 captions: S = 128,
// CLIP ViT-B/32: T = 50), forward and backward, on MFMA 16x16x32 bf16.  Longer sequences (S = 256 captions,
// ViT-L/14's 257 tokens) take the key-tiled kernels of attention_tiled.hip through the same two entry points.
//
// One workgroup (8 waves, two per SIMD) owns one (batch, head): the whole S x S score tile lives on chip,
// so nothing is summed across workgroups (no dQ atomics) and the scores never reach HBM.
// Wave w owns query rows 16w..16w+15.  Scores are computed TRANSPOSED (S^t = K Q^t) so a
// lane holds 4 consecutive keys of one query: the softmax row-reduction is in-lane plus two
// shuffles, and P is written to LDS with 8-byte stores.  Operands that are strided along the
// contraction (V in P.V, and P^t / dS^t / dO / Q / K in the backward products) are read with
// ds_read_b64_tr_b16, so every tile is staged once in its natural [row][64] layout.
// Causal blocks above the diagonal are skipped (wave-uniform loop bounds).
#include <stdlib.h>

#include "common.h"

using namespace pgca;

namespace {

constexpr int NWAVE = 8;      // waves per workgroup: two per SIMD, each owning RW query (or key) rows
constexpr int NT = 64 * NWAVE;
constexpr int RW = 128 / NWAVE;
constexpr int NI = RW / 16;   // 16-row MFMA tiles per wave
constexpr int SP = 128;       // padded sequence tile
constexpr int DH = 64;        // head dim
constexpr int QS = 144;       // byte stride of a [.][64] bf16 row (128 + 16 pad)
constexpr int PS = 272;       // byte stride of a [.][128] bf16 row (256 + 16 pad)
constexpr int TILE_QKV = SP * QS;  // 18432
constexpr int TILE_P = SP * PS;    // 34816

// Stage a [S][64] bf16 head slice (row stride `ld` elements in global) into LDS, zero rows >= S.
__device__ __forceinline__ void stage_head(unsigned char* lds, const bf16_t* g, int ld, int S, int t) {
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    const int idx = t + NT * i;  // 1024 chunks of 16 B
    const int row = idx >> 3, c = idx & 7;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (row < S) v = *reinterpret_cast<const u32x4*>(g + (size_t)row * ld + c * 8);
    *reinterpret_cast<u32x4*>(lds + row * QS + c * 16) = v;
  }
}

// K-contiguous fragment (rows = MFMA row/col index, 64-deep k) from a [.][64] image.
__device__ __forceinline__ bf16x8 frag_rows64(const unsigned char* lds, int row0, int kk, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 15)) * QS + kk * 64 + (lane >> 4) * 16);
}
// K-contiguous fragment from a [.][128] image (P / dS rows), k-step ks of 32.
__device__ __forceinline__ bf16x8 frag_rows128(const unsigned char* lds, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 15)) * PS + ks * 64 + (lane >> 4) * 16);
}
// K-strided fragment: image T[k][.] with byte row stride `stride`; k rows k0..k0+31, 16 columns at col0.
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* lds, int stride, int k0, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a0 = lds + (k0 + 8 * g + q) * stride + (col0 + 4 * p) * 2;
  return tr_frag(a0, a0 + 4 * stride);
}

__device__ __forceinline__ bool key_ok(int key, int q, int S, int causal, const unsigned char* kms) {
  return key < S && (!causal || key <= q) && kms[key];
}

// ------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(NT, 4) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ kmask,
                                                          int S, int heads, int causal, bf16_t* __restrict__ out,
                                                          float* __restrict__ lse_o, Drop drop) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + TILE_QKV;
  unsigned char* Ps = smem + 2 * TILE_QKV;
  unsigned char* kms = smem + 2 * TILE_QKV + TILE_P;  // [128] bytes

  const int h = blockIdx.x, b = blockIdx.y;
  const int H = heads * DH, ld = 3 * H;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * DH;

  stage_head(Ks, base + H, ld, S, t);
  stage_head(Vs, base + 2 * H, ld, S, t);
  if (t < SP) kms[t] = (t < S && (!kmask || kmask[b * S + t] != 0)) ? 1 : 0;

  // Q fragments straight from global: rows 32w + ni*16 + (lane&15)
  bf16x8 fq[NI][2];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int q = RW * w + ni * 16 + (lane & 15);
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (q < S) v = *reinterpret_cast<const u32x4*>(base + (size_t)q * ld + kk * 32 + (lane >> 4) * 8);
      fq[ni][kk] = __builtin_bit_cast(bf16x8, v);
    }
  __syncthreads();

  const int ntile = causal ? min((w + 1) * NI, (S + 15) >> 4) : ((S + 15) >> 4);  // key tiles this wave needs
  const float scale = 0.125f;

  f32x4 acc[8][NI];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (mi < ntile) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 fk = frag_rows64(Ks, mi * 16, kk, lane);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[ni][kk], acc[mi][ni], 0, 0, 0);
      }
    }
  }

  // softmax over keys for each query column (ni, lane&15); keys live on (mi, lane>>4, r)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = RW * w + ni * 16 + (lane & 15);
    float mx = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = mi * 16 + (lane >> 4) * 4 + r;
        const float s = acc[mi][ni][r] * scale;
        acc[mi][ni][r] = (mi < ntile && key_ok(key, q, S, causal, kms)) ? s : -INFINITY;
        mx = fmaxf(mx, acc[mi][ni][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = acc[mi][ni][r] > -INFINITY ? __expf(acc[mi][ni][r] - mx) : 0.f;
        acc[mi][ni][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    if (lane < 16 && q < S && lse_o) lse_o[((size_t)b * heads + h) * S + q] = mx + __logf(sum);
    // attention-probability dropout (GPT-2 attn_dropout, modeling_gpt2.py:66): element (b, h, q, key)
    const unsigned dbase = (((unsigned)b * heads + h) * S + q) * S;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = acc[mi][ni][r] * inv;
        if (drop.on()) pv[r] *= drop.mul(dbase + mi * 16 + (lane >> 4) * 4 + r);
      }
      u32x2 pk;
      pk[0] = pack2(pv[0], pv[1]);
      pk[1] = pack2(pv[2], pv[3]);
      *reinterpret_cast<u32x2*>(Ps + q * PS + (mi * 16 + (lane >> 4) * 4) * 2) = pk;
    }
  }
  __syncthreads();

  // O = P V : rows 32w + mt*16.., cols nt*16.., contraction over keys in steps of 32
  f32x4 o[NI][4];
#pragma unroll
  for (int mt = 0; mt < NI; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) o[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nks = (ntile + 1) >> 1;
  for (int ks = 0; ks < nks; ++ks) {
    bf16x8 fp[NI], fv[4];
#pragma unroll
    for (int mt = 0; mt < NI; ++mt) fp[mt] = frag_rows128(Ps, RW * w + mt * 16, ks, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fv[nt] = frag_tr(Vs, QS, ks * 32, nt * 16, lane);
#pragma unroll
    for (int mt = 0; mt < NI; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        o[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp[mt], fv[nt], o[mt][nt], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < NI; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = RW * w + mt * 16 + (lane >> 4) * 4 + r;
      if (q < S) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          out[((size_t)b * S + q) * H + h * DH + nt * 16 + (lane & 15)] = f2bf(o[mt][nt][r]);
      }
    }
}

// ------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(NT, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ O,
                                                          const bf16_t* __restrict__ dO, const float* __restrict__ lse_i,
                                                          const int* __restrict__ kmask, int S, int heads, int causal,
                                                          bf16_t* __restrict__ dqkv, Drop drop) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem;
  unsigned char* Ks = smem + TILE_QKV;
  unsigned char* Vs = smem + 2 * TILE_QKV;
  unsigned char* dOs = smem + 3 * TILE_QKV;
  unsigned char* Ps = smem + 4 * TILE_QKV;
  unsigned char* dSs = Ps + TILE_P;
  float* lses = reinterpret_cast<float*>(dSs + TILE_P);  // [128]
  float* dels = lses + SP;                               // [128]
  unsigned char* kms = reinterpret_cast<unsigned char*>(dels + SP);

  const int h = blockIdx.x, b = blockIdx.y;
  const int H = heads * DH, ld = 3 * H;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * DH;
  const bf16_t* obase = O + (size_t)b * S * H + h * DH;
  const bf16_t* dobase = dO + (size_t)b * S * H + h * DH;

  stage_head(Qs, base, ld, S, t);
  stage_head(Ks, base + H, ld, S, t);
  stage_head(Vs, base + 2 * H, ld, S, t);
  stage_head(dOs, dobase, H, S, t);
  if (t < SP) {
    kms[t] = (t < S && (!kmask || kmask[b * S + t] != 0)) ? 1 : 0;
    lses[t] = t < S ? lse_i[((size_t)b * heads + h) * S + t] : 0.f;
  }
  // delta[q] = sum_d dO[q,d] * O[q,d]; thread handles chunk (row = idx>>3, c = idx&7); 8 lanes per row
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    const int idx = t + NT * i;
    const int row = idx >> 3, c = idx & 7;
    float d = 0.f;
    if (row < S) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(obase + (size_t)row * H + c * 8);
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(dobase + (size_t)row * H + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) d += (float)a[e] * (float)g[e];
    }
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 4);
    if (c == 0) dels[row] = d;
  }
  __syncthreads();

  const int nkt = (S + 15) >> 4;
  const int ntile = causal ? min((w + 1) * NI, nkt) : nkt;
  const float scale = 0.125f;

  // Phase 1: P and dS (as [q][key]) for this wave's 32 query rows.
  bf16x8 fq[NI][2], fdo[NI][2];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      fq[ni][kk] = frag_rows64(Qs, RW * w + ni * 16, kk, lane);
      fdo[ni][kk] = frag_rows64(dOs, RW * w + ni * 16, kk, lane);
    }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    f32x4 s[NI], dp[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) s[ni] = dp[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (mi < ntile) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 fk = frag_rows64(Ks, mi * 16, kk, lane);
        const bf16x8 fv = frag_rows64(Vs, mi * 16, kk, lane);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          s[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[ni][kk], s[ni], 0, 0, 0);
          dp[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fdo[ni][kk], dp[ni], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int q = RW * w + ni * 16 + (lane & 15);
      const float l = lses[q], dl = dels[q];
      const unsigned dbase = (((unsigned)b * heads + h) * S + q) * S;
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = mi * 16 + (lane >> 4) * 4 + r;
        const bool ok = mi < ntile && q < S && key_ok(key, q, S, causal, kms);
        const float pu = ok ? __expf(s[ni][r] * scale - l) : 0.f;       // undropped probability
        const float m = drop.on() ? drop.mul(dbase + key) : 1.f;          // replayed dropout multiplier
        ds[r] = pu * (dp[ni][r] * m - dl) * scale;                        // dP = dP_dropped * m
        p[r] = pu * m;                                                    // dV uses the dropped probabilities
      }
      u32x2 pk, dk;
      pk[0] = pack2(p[0], p[1]);
      pk[1] = pack2(p[2], p[3]);
      dk[0] = pack2(ds[0], ds[1]);
      dk[1] = pack2(ds[2], ds[3]);
      const int off = q * PS + (mi * 16 + (lane >> 4) * 4) * 2;
      *reinterpret_cast<u32x2*>(Ps + off) = pk;
      *reinterpret_cast<u32x2*>(dSs + off) = dk;
    }
  }
  __syncthreads();

  // Phase 2: dV = P^t dO, dK = dS^t Q for key rows 32w..32w+31; contraction over queries.
  {
    f32x4 dv[NI][4], dk[NI][4];
#pragma unroll
    for (int mt = 0; mt < NI; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dv[mt][nt] = dk[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nqs = (S + 31) >> 5;
    for (int qs = causal ? ((RW * w) >> 5) : 0; qs < nqs; ++qs) {
      bf16x8 fpt[NI], fdst[NI], fdo2[4], fq2[4];
#pragma unroll
      for (int mt = 0; mt < NI; ++mt) {
        fpt[mt] = frag_tr(Ps, PS, qs * 32, RW * w + mt * 16, lane);
        fdst[mt] = frag_tr(dSs, PS, qs * 32, RW * w + mt * 16, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        fdo2[nt] = frag_tr(dOs, QS, qs * 32, nt * 16, lane);
        fq2[nt] = frag_tr(Qs, QS, qs * 32, nt * 16, lane);
      }
#pragma unroll
      for (int mt = 0; mt < NI; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          dv[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fpt[mt], fdo2[nt], dv[mt][nt], 0, 0, 0);
          dk[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdst[mt], fq2[nt], dk[mt][nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < NI; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = RW * w + mt * 16 + (lane >> 4) * 4 + r;
        if (key < S) {
          bf16_t* row = dqkv + ((size_t)b * S + key) * ld + h * DH;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            row[H + nt * 16 + (lane & 15)] = f2bf(dk[mt][nt][r]);
            row[2 * H + nt * 16 + (lane & 15)] = f2bf(dv[mt][nt][r]);
          }
        }
      }
  }

  // Phase 3: dQ = dS K for query rows 32w..32w+31; contraction over keys.
  {
    f32x4 dq[NI][4];
#pragma unroll
    for (int mt = 0; mt < NI; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dq[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = (ntile + 1) >> 1;
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 fds[NI], fk2[4];
#pragma unroll
      for (int mt = 0; mt < NI; ++mt) fds[mt] = frag_rows128(dSs, RW * w + mt * 16, ks, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fk2[nt] = frag_tr(Ks, QS, ks * 32, nt * 16, lane);
#pragma unroll
      for (int mt = 0; mt < NI; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          dq[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fds[mt], fk2[nt], dq[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < NI; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = RW * w + mt * 16 + (lane >> 4) * 4 + r;
        if (q < S) {
          bf16_t* row = dqkv + ((size_t)b * S + q) * ld + h * DH;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) row[nt * 16 + (lane & 15)] = f2bf(dq[mt][nt][r]);
        }
      }
  }
}

constexpr size_t FWD_LDS = 2 * TILE_QKV + TILE_P + SP;
constexpr size_t BWD_LDS = 4 * TILE_QKV + 2 * TILE_P + 2 * SP * sizeof(float) + SP;

int ensure_lds_attr() {
  // function-local static: initialised exactly once, thread-safe (C++11)
  static const bool ok =
      hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD_LDS) ==
          hipSuccess &&
      hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS) ==
          hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    set_error("attention: cannot raise dynamic LDS limit");
    return PGCA_ERR_LAUNCH;
  }
  return PGCA_OK;
}

// PGCA_ATTN_TILED=1 (read once) routes S <= 128 through the key-tiled kernels too (A/B timing, parity tests)
bool force_tiled() {
  static const bool v = [] {
    const char* e = getenv("PGCA_ATTN_TILED");
    return e && atoi(e) != 0;
  }();
  return v;
}

}  // namespace

namespace pgca {
int attention_fwd_tiled(const void* qkv, const int32_t* key_mask, int B, int S, int heads, int causal, void* out,
                        float* lse, uint32_t drop_seed, uint32_t drop_threshold, float drop_scale, void* stream);
int attention_bwd_tiled(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_mask,
                        int B, int S, int heads, int causal, void* dqkv, uint32_t drop_seed, uint32_t drop_threshold,
                        float drop_scale, void* stream);
}  // namespace pgca

extern "C" int pgca_attention_fwd(const void* qkv, const int32_t* key_mask, int32_t B, int32_t S, int32_t heads,
                                  int32_t causal, void* out, float* lse, uint32_t drop_seed, uint32_t drop_threshold,
                                  float drop_scale, void* stream) {
  if (!qkv || !out || B <= 0 || S <= 0 || heads <= 0 || B > 65535) {
    set_error("pgca_attention_fwd: bad arguments (B=%d S=%d heads=%d)", B, S, heads);
    return PGCA_ERR_INVALID;
  }
  if (S > SP || force_tiled())
    return attention_fwd_tiled(qkv, key_mask, B, S, heads, causal, out, lse, drop_seed, drop_threshold, drop_scale,
                               stream);
  if (ensure_lds_attr()) return PGCA_ERR_LAUNCH;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(heads, B), dim3(NT), FWD_LDS, (hipStream_t)stream, (const bf16_t*)qkv,
                     key_mask, S, heads, causal, (bf16_t*)out, lse, Drop{drop_seed, drop_threshold, drop_scale});
  return check_launch("pgca_attention_fwd");
}

extern "C" int pgca_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                  const int32_t* key_mask, int32_t B, int32_t S, int32_t heads, int32_t causal,
                                  void* dqkv, uint32_t drop_seed, uint32_t drop_threshold, float drop_scale,
                                  void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || S <= 0 || S > PGCA_ATTN_MAX_S || heads <= 0 || B > 65535) {
    set_error("pgca_attention_bwd: bad arguments (B=%d S=%d heads=%d; S must be <= %d)", B, S, heads, PGCA_ATTN_MAX_S);
    return PGCA_ERR_INVALID;
  }
  if (S > SP || force_tiled())
    return attention_bwd_tiled(qkv, out, dout, lse, key_mask, B, S, heads, causal, dqkv, drop_seed, drop_threshold,
                               drop_scale, stream);
  if (ensure_lds_attr()) return PGCA_ERR_LAUNCH;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(heads, B), dim3(NT), BWD_LDS, (hipStream_t)stream, (const bf16_t*)qkv,
                     (const bf16_t*)out, (const bf16_t*)dout, lse, key_mask, S, heads, causal, (bf16_t*)dqkv,
                     Drop{drop_seed, drop_threshold, drop_scale});
  return check_launch("pgca_attention_bwd");
}
