// Key-tiled (flash-style) multi-head self-attention for head_dim 64 and every sequence length of the hot path:
// GPT-2 captions at S = 128 (one tile) and S = 256 (configs C4/C5, causal + key padding + replayed probability
// dropout), CLIP ViT-B/32 at T = 50 and ViT-L/14 at T = 257 (no mask).  Entry points: attention.hip.
//
// Forward: one workgroup (8 waves) per (head, batch, 128-query block); wave w owns 16 queries and sweeps the key
// tiles with an online softmax.  Scores are computed TRANSPOSED (S^t = K Q^t) so a lane holds 4 keys of ONE query per
// MFMA tile: the row maximum / sum are in-lane plus two shuffles, and - the point of the layout - the exponentiated
// accumulator IS the B operand of the next product O^t += V^t P^t (lane (g, n = query) holds the k-slots 8g..8g+7;
// the k-slot <-> key map is the bijection {32s+4g+j, 32s+16+4g+j} and the V^t fragment is read with the same
// map by ds_read_b64_tr_b16), so the probabilities never touch LDS.  O^t rows live on the same lanes as the softmax
// statistics, so the running rescale is a per-lane scalar.  O leaves through a wave-private LDS transpose as 16-byte
// row-contiguous stores.
//
// Backward: one workgroup per (head, batch) sweeps (key block, query block) pairs.  S and dP are computed with the KEY
// on the lane, so P and dS accumulators feed dV^t += dO^t P and dK^t += Q^t dS directly as B operands; only dS crosses
// LDS (once, transposed, 8-byte stores) for dQ += dS K, whose accumulators stay in registers for every query block
// (template NQB): no atomics, nothing summed across workgroups, bitwise reproducible.
#include "common.h"

using namespace pgca;

namespace {

constexpr int TNT = 512;       // 8 waves
constexpr int TB = 128;        // rows per query / key block
constexpr int DH = 64;
constexpr int QS = 144;        // byte stride of a [.][64] bf16 row (128 + 16 pad)
constexpr int PS = 272;        // byte stride of a [.][128] bf16 row (256 + 16 pad)
constexpr int TILE_QKV = TB * QS;  // 18432
constexpr int TILE_P = TB * PS;    // 34816

// rows row0 .. row0+127 of a [S][64] head slice (row stride ld elements) -> LDS image, zero rows >= S
__device__ __forceinline__ void stage_rows(unsigned char* lds, const bf16_t* g, int ld, int row0, int S, int t) {
#pragma unroll
  for (int i = 0; i < 1024 / TNT; ++i) {
    const int idx = t + TNT * i;
    const int row = idx >> 3, c = idx & 7;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (row0 + row < S) v = *reinterpret_cast<const u32x4*>(g + (size_t)(row0 + row) * ld + c * 8);
    *reinterpret_cast<u32x4*>(lds + row * QS + c * 16) = v;
  }
}

// the same in two halves, so that the global loads of several tiles can be in flight together before any of them is waited for
struct StageRegs {
  u32x4 v[1024 / TNT];
};
__device__ __forceinline__ void stage_load(StageRegs& r, const bf16_t* g, int ld, int row0, int S, int t) {
#pragma unroll
  for (int i = 0; i < 1024 / TNT; ++i) {
    const int idx = t + TNT * i;
    const int row = idx >> 3, c = idx & 7;
    r.v[i] = (u32x4){0u, 0u, 0u, 0u};
    if (row0 + row < S) r.v[i] = *reinterpret_cast<const u32x4*>(g + (size_t)(row0 + row) * ld + c * 8);
  }
}
__device__ __forceinline__ void stage_store(const StageRegs& r, unsigned char* lds, int t) {
#pragma unroll
  for (int i = 0; i < 1024 / TNT; ++i) {
    const int idx = t + TNT * i;
    *reinterpret_cast<u32x4*>(lds + (idx >> 3) * QS + (idx & 7) * 16) = r.v[i];
  }
}

// MFMA operand whose 16 rows (A) / 16 columns (B) are image rows row0..row0+15 and whose k is contiguous (64-deep image)
__device__ __forceinline__ bf16x8 frag_rows64(const unsigned char* lds, int row0, int kk, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 15)) * QS + kk * 64 + (lane >> 4) * 16);
}
// k-strided fragment, standard k map: k-slot (g, j) <-> image row k0 + 8g + j; 16 columns at col0
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* lds, int stride, int k0, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a0 = lds + (k0 + 8 * g + q) * stride + (col0 + 4 * p) * 2;
  return tr_frag(a0, a0 + 4 * stride);
}
// k-strided fragment, ACCUMULATOR k map: k-slot (g, j < 4) <-> row k0 + 4g + j, (g, j >= 4) <-> row k0 + 16 + 4g + j - 4:
// the order in which two stacked 16x16 accumulator tiles present their rows to a lane
__device__ __forceinline__ bf16x8 frag_tr_acc(const unsigned char* lds, int stride, int k0, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a0 = lds + (k0 + 4 * g + q) * stride + (col0 + 4 * p) * 2;
  return tr_frag(a0, a0 + 16 * stride);
}
// two accumulator tiles -> the bf16 B operand (same k map as frag_tr_acc)
__device__ __forceinline__ bf16x8 pack_acc(const float (&lo)[4], const float (&hi)[4]) {
  u32x4 v;
  v[0] = pack2(lo[0], lo[1]);
  v[1] = pack2(lo[2], lo[3]);
  v[2] = pack2(hi[0], hi[1]);
  v[3] = pack2(hi[2], hi[3]);
  return __builtin_bit_cast(bf16x8, v);
}

// a wave's 16 x 64 bf16 tile, staged in its private LDS rows (row stride `stride` bytes, 128 bytes used per row),
// leaves as 16-byte stores: 8 lanes cover one 128-byte row
__device__ __forceinline__ void store_rows16(const unsigned char* stage, int stride, bf16_t* gbase, size_t grow_stride,
                                             int row0, int nrows_valid, int lane) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int cid = lane + 64 * it;
    const int row = cid >> 3, c = cid & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * stride + c * 16);
    if (row < nrows_valid) *reinterpret_cast<u32x4*>(gbase + (size_t)(row0 + row) * grow_stride + c * 8) = v;
  }
}

// ------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(TNT, 4) void attn_fwd_tiled_kernel(const bf16_t* __restrict__ qkv,
                                                                const int* __restrict__ kmask, int S, int heads,
                                                                int causal, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse_o, Drop drop,
                                                                const int* __restrict__ cu, int Sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + TILE_QKV;
  unsigned char* kms = smem + 2 * TILE_QKV;  // [128] bytes

  // Sp: the PADDED sequence length (geometry of key_mask, lse and the dropout index).  Packed rows (cu != NULL): this
  // sequence's tokens are rows cu[b] .. cu[b+1]-1 and S is their count; otherwise rows b*Sp .. and S == Sp.
  const int h = blockIdx.x, b = blockIdx.y, qb = blockIdx.z;
  const int row0 = cu ? cu[b] : b * Sp;
  if (cu) S = min(S, cu[b + 1] - row0);   // (a KV-cache launch gives cu the cache stride and S the filled length)
  if (qb * TB >= S) return;  // block-uniform: a short (or empty) sequence has no such query block
  const int H = heads * DH, ld = 3 * H;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, i16 = lane & 15;
  const bf16_t* base = qkv + (size_t)row0 * ld + h * DH;

  const int qw = qb * TB + 16 * w;  // this wave's first query
  const int q = qw + i16;
  const bool wave_on = qw < S;      // wave-uniform

  bf16x8 fq[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (q < S) v = *reinterpret_cast<const u32x4*>(base + (size_t)q * ld + kk * 32 + g * 8);
    fq[kk] = __builtin_bit_cast(bf16x8, v);
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 oT[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float scale = 0.125f;
  const int nkt = causal ? qb + 1 : (S + TB - 1) / TB;
  const unsigned dbase = (((unsigned)b * heads + h) * Sp + q) * Sp;

  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * TB;
    __syncthreads();  // every wave is done with the previous key tile
    stage_rows(Ks, base + H, ld, k0, S, t);
    stage_rows(Vs, base + 2 * H, ld, k0, S, t);
    if (t < TB) kms[t] = (k0 + t < S && (!kmask || kmask[b * Sp + k0 + t] != 0)) ? 1 : 0;
    __syncthreads();
    if (!wave_on) continue;

    const int nvalid = min(8, (S - k0 + 15) >> 4);
    const int ntile = (causal && kt == qb) ? min(w + 1, nvalid) : nvalid;  // 16-key tiles this wave needs

    f32x4 acc[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      acc[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (mi < ntile) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows64(Ks, mi * 16, kk, lane), fq[kk], acc[mi], 0, 0, 0);
      }
    }
    // scale + mask; running maximum of this query (keys live on (mi, g, r))
    float mx = m_run;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      // the 4 validity bytes of this lane's keys of tile mi in one LDS word (kms already folds key < S)
      const unsigned kv = *reinterpret_cast<const unsigned*>(kms + mi * 16 + g * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + mi * 16 + g * 4 + r;
        const bool ok = mi < ntile && ((kv >> (8 * r)) & 1u) && (!causal || key <= q);
        const float s = ok ? acc[mi][r] * scale : -INFINITY;
        acc[mi][r] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float alpha = mx > -INFINITY ? __expf(m_run - mx) : 1.f;  // exp(-inf) = 0 when this is the first live tile
    m_run = mx;
    float psum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      // attention-probability dropout (GPT-2 attn_dropout, modeling_gpt2.py:66) acts on the NORMALISED probability:
      // the mask multiplies the numerator only, the row sum stays undropped
      float dm[4] = {1.f, 1.f, 1.f, 1.f};
      if (drop.on() && mi < ntile) drop.mul4(dbase + (unsigned)(k0 + mi * 16 + g * 4), dm);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = acc[mi][r] > -INFINITY ? __expf(acc[mi][r] - mx) : 0.f;
        psum += p;
        acc[mi][r] = p * dm[r];
      }
    }
    l_run = l_run * alpha + psum;  // per-lane partial; the four g-lanes of a query are summed once at the end
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oT[dt][r] *= alpha;
    // O^t[d][q] += V^t[d][key] P^t[key][q], 32 keys per step
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (2 * s < ntile) {
        const float lo[4] = {acc[2 * s][0], acc[2 * s][1], acc[2 * s][2], acc[2 * s][3]};
        const float hi[4] = {acc[2 * s + 1][0], acc[2 * s + 1][1], acc[2 * s + 1][2], acc[2 * s + 1][3]};
        const bf16x8 fp = pack_acc(lo, hi);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          oT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_acc(Vs, QS, s * 32, dt * 16, lane), fp, oT[dt], 0, 0, 0);
      }
    }
  }

  float l = l_run;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (lane < 16 && q < S && lse_o) lse_o[((size_t)b * heads + h) * Sp + q] = m_run + __logf(l);

  __syncthreads();  // Ks is free: it becomes the output staging (wave w uses rows 16w..16w+15 only)
  unsigned char* stage = Ks + (16 * w) * QS;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    u32x2 pk;
    pk[0] = pack2(oT[dt][0] * inv, oT[dt][1] * inv);
    pk[1] = pack2(oT[dt][2] * inv, oT[dt][3] * inv);
    *reinterpret_cast<u32x2*>(stage + i16 * QS + (dt * 16 + g * 4) * 2) = pk;
  }
  if (wave_on) store_rows16(stage, QS, out + (size_t)row0 * H + h * DH, H, qw, min(16, S - qw), lane);
}

// ------------------------------------------------------------------------------------ forward, S <= 128
// One-tile sequences (every training shape, the ViT-B towers, the decode cache up to 128 tokens): a FOUR-wave workgroup per
// (sequence, head).  K and V are staged once; each wave then walks up to two 16-query blocks (rows 16w.. and 64+16w..) one
// after the other.  Against the eight-wave kernel above this (i) puts four workgroups on a CU instead of two at the same
// 128 registers - twice as many staging loads in flight for a kernel that mostly waits for memory -, (ii) leaves no wave
// idle when the sequence is short (packed captions average 72 tokens: 5 of 8 waves had queries, and a sequence of <= 64
// tokens now occupies 4 wave slots instead of 8), (iii) needs no online-softmax rescale: one key tile, one maximum.
constexpr int FNT = 256;
__global__ __launch_bounds__(FNT, 4) void attn_fwd_one_kernel(const bf16_t* __restrict__ qkv,
                                                              const int* __restrict__ kmask, int S, int heads, int causal,
                                                              bf16_t* __restrict__ out, float* __restrict__ lse_o,
                                                              Drop drop, const int* __restrict__ cu, int Sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + TILE_QKV;
  unsigned char* kms = smem + 2 * TILE_QKV;  // [128] bytes
  const int h = blockIdx.x, b = blockIdx.y;
  const int row0 = cu ? cu[b] : b * Sp;
  if (cu) S = min(S, cu[b + 1] - row0);
  if (S <= 0) return;
  const int H = heads * DH, ld = 3 * H;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, i16 = lane & 15;
  const bf16_t* base = qkv + (size_t)row0 * ld + h * DH;

  // all of this workgroup's loads are requested before anything waits: K, V (4 + 4 chunks per thread) and both Q blocks
  u32x4 rk[4], rv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + FNT * i, row = idx >> 3, c = idx & 7;
    rk[i] = rv[i] = (u32x4){0u, 0u, 0u, 0u};
    if (row < S) {
      rk[i] = *reinterpret_cast<const u32x4*>(base + H + (size_t)row * ld + c * 8);
      rv[i] = *reinterpret_cast<const u32x4*>(base + 2 * H + (size_t)row * ld + c * 8);
    }
  }
  bf16x8 fq[2][2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int q = 64 * ps + 16 * w + i16;
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (q < S) v = *reinterpret_cast<const u32x4*>(base + (size_t)q * ld + kk * 32 + g * 8);
      fq[ps][kk] = __builtin_bit_cast(bf16x8, v);
    }
  if (t < TB) kms[t] = (t < S && (!kmask || kmask[b * Sp + t] != 0)) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + FNT * i, row = idx >> 3, c = idx & 7;
    *reinterpret_cast<u32x4*>(Ks + row * QS + c * 16) = rk[i];
    *reinterpret_cast<u32x4*>(Vs + row * QS + c * 16) = rv[i];
  }
  __syncthreads();

  const float scale = 0.125f;
  const int nvalid = min(8, (S + 15) >> 4);
  u32x2 opk[2][4];  // the two blocks' outputs, packed bf16, until K's space can stage them
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) opk[ps][dt] = (u32x2){0u, 0u};
    const int qw = 64 * ps + 16 * w;  // this wave's first query of the block
    if (qw < S) {                      // wave-uniform
      const int q = qw + i16;
      const int ntile = causal ? min((qw >> 4) + 1, nvalid) : nvalid;  // 16-key tiles this block needs
      f32x4 acc[8];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        acc[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (mi < ntile) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows64(Ks, mi * 16, kk, lane), fq[ps][kk], acc[mi], 0, 0,
                                                              0);
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const unsigned kv = *reinterpret_cast<const unsigned*>(kms + mi * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = mi * 16 + g * 4 + r;
          const bool ok = mi < ntile && ((kv >> (8 * r)) & 1u) && (!causal || key <= q);
          const float sc = ok ? acc[mi][r] * scale : -INFINITY;
          acc[mi][r] = sc;
          mx = fmaxf(mx, sc);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const unsigned dbase = (((unsigned)b * heads + h) * Sp + q) * Sp;
      float l = 0.f;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        float dm[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop.on() && mi < ntile) drop.mul4(dbase + (unsigned)(mi * 16 + g * 4), dm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = acc[mi][r] > -INFINITY ? __expf(acc[mi][r] - mx) : 0.f;
          l += pr;
          acc[mi][r] = pr * dm[r];
        }
      }
      f32x4 oT[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (2 * s4 < ntile) {
          const float lo[4] = {acc[2 * s4][0], acc[2 * s4][1], acc[2 * s4][2], acc[2 * s4][3]};
          const float hi[4] = {acc[2 * s4 + 1][0], acc[2 * s4 + 1][1], acc[2 * s4 + 1][2], acc[2 * s4 + 1][3]};
          const bf16x8 fp = pack_acc(lo, hi);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            oT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_acc(Vs, QS, s4 * 32, dt * 16, lane), fp, oT[dt], 0, 0, 0);
        }
      }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv = l > 0.f ? 1.f / l : 0.f;
      if (lane < 16 && q < S && lse_o) lse_o[((size_t)b * heads + h) * Sp + q] = mx + __logf(l);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        opk[ps][dt][0] = pack2(oT[dt][0] * inv, oT[dt][1] * inv);
        opk[ps][dt][1] = pack2(oT[dt][2] * inv, oT[dt][3] * inv);
      }
    }
  }
  __syncthreads();  // every wave is done with K: its rows become the output staging (block ps of wave w: rows 64ps + 16w ..)
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int qw = 64 * ps + 16 * w;
    if (qw < S) {
      unsigned char* stage = Ks + qw * QS;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<u32x2*>(stage + i16 * QS + (dt * 16 + g * 4) * 2) = opk[ps][dt];
      store_rows16(stage, QS, out + (size_t)row0 * H + h * DH, H, qw, min(16, S - qw), lane);
    }
  }
}

// ------------------------------------------------------------------------------------ backward
// ALIAS (one-block sequences only): the dS^t image lives where V and Q were - both are dead once phase A is over - so
// a workgroup needs 75 KiB of LDS instead of 109 and two of them share a CU (registers capped at 128 for that).
template <int NQB, bool ALIAS>
__global__ __launch_bounds__(TNT, ALIAS ? 4 : 2) void attn_bwd_tiled_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ O,
                                                                const bf16_t* __restrict__ dO,
                                                                const float* __restrict__ lse_i,
                                                                const int* __restrict__ kmask, int S, int heads,
                                                                int causal, bf16_t* __restrict__ dqkv, Drop drop,
                                                                const int* __restrict__ cu, int Sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + TILE_QKV;
  unsigned char* Qs = smem + 2 * TILE_QKV;
  unsigned char* dOs = smem + 3 * TILE_QKV;
  static_assert(!ALIAS || (NQB == 1 && TILE_P <= 2 * TILE_QKV), "aliasing needs a single (key, query) pair");
  unsigned char* dSs = ALIAS ? Vs : smem + 4 * TILE_QKV;           // dS^t image [128 keys][128 queries]
  float* lses = reinterpret_cast<float*>(smem + 4 * TILE_QKV + (ALIAS ? 0 : TILE_P));  // [NQB * 128]
  float* dels = lses + NQB * TB;                                   // [NQB * 128]
  unsigned char* kms = reinterpret_cast<unsigned char*>(dels + NQB * TB);  // [128]

  const int b = blockIdx.y;
  const int row0 = cu ? cu[b] : b * Sp;  // packed rows: see the forward
  if (cu) S = min(S, cu[b + 1] - row0);
  if (S <= 0) return;
  const int H = heads * DH, ld = 3 * H;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, i16 = lane & 15;

  // A workgroup's memory latency is hidden by at most one other workgroup on its CU (none when S > 128): the first
  // (key block, query block) pair - K, V, Q, dO tiles - is REQUESTED together with the row constants' inputs (O rows for
  // delta, lse) into registers and stored to LDS afterwards: one HBM round trip in front of the first MFMA instead of
  // three.  With packed sequences (mean length 72 of 128) most workgroups have exactly one pair.
  constexpr int NLSE = (NQB * TB + TNT - 1) / TNT;
  const int h = blockIdx.x;
  const bf16_t* base = qkv + (size_t)row0 * ld + h * DH;
  const bf16_t* obase = O + (size_t)row0 * H + h * DH;
  const bf16_t* dobase = dO + (size_t)row0 * H + h * DH;
  bf16_t* dbase_g = dqkv + (size_t)row0 * ld + h * DH;
  StageRegs pk, pv, pq, pdo, po;
  float plse[NLSE];
  stage_load(pk, base + H, ld, 0, S, t);
  stage_load(pv, base + 2 * H, ld, 0, S, t);
  stage_load(pq, base, ld, 0, S, t);
  stage_load(pdo, dobase, H, 0, S, t);
  stage_load(po, obase, H, 0, S, t);
#pragma unroll
  for (int j = 0; j < NLSE; ++j) {
    const int i = t + TNT * j;
    plse[j] = i < S ? lse_i[((size_t)b * heads + h) * Sp + i] : 0.f;
  }

  // row constants of every query: lse and delta[q] = sum_d dO[q,d] O[q,d] (8 lanes per row).  Rows of the first query
  // block come from the requested registers (stage_load's (row, chunk) map is this loop's), later blocks from memory.
#pragma unroll
  for (int i = 0; i < 1024 / TNT; ++i) {
    const int idx = t + TNT * i;
    const int row = idx >> 3, c = idx & 7;
    const bf16x8 a = __builtin_bit_cast(bf16x8, po.v[i]);
    const bf16x8 gg = __builtin_bit_cast(bf16x8, pdo.v[i]);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += (float)a[e] * (float)gg[e];
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 4);
    if (c == 0) dels[row] = d;
  }
  for (int idx = t + 1024; idx < NQB * TB * 8; idx += TNT) {
    const int row = idx >> 3, c = idx & 7;
    float d = 0.f;
    if (row < S) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(obase + (size_t)row * H + c * 8);
      const bf16x8 gg = *reinterpret_cast<const bf16x8*>(dobase + (size_t)row * H + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) d += (float)a[e] * (float)gg[e];
    }
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 4);
    if (c == 0) dels[row] = d;
  }
#pragma unroll
  for (int j = 0; j < NLSE; ++j)
    if (t + TNT * j < NQB * TB) lses[t + TNT * j] = plse[j];

  f32x4 dq[NQB][4];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dq[qb][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float scale = 0.125f;
  const int nblk = NQB == 1 ? 1 : (S + TB - 1) / TB;  // key blocks == query blocks (<= NQB; S > 0 here)
  unsigned char* mine = dSs + (16 * w) * PS;  // this wave's own 16 rows of the dS^t image (also its store staging)

  for (int kb = 0; kb < nblk; ++kb) {
    const int k0 = kb * TB;
    // (the barrier that closes the previous pair protects Ks / Vs / kms)
    if (kb == 0) {   // requested at the top; the first query block (0 in both mask modes) comes with them
      stage_store(pk, Ks, t);
      stage_store(pv, Vs, t);
      stage_store(pq, Qs, t);
      stage_store(pdo, dOs, t);
    } else {
      stage_rows(Ks, base + H, ld, k0, S, t);
      stage_rows(Vs, base + 2 * H, ld, k0, S, t);
    }
    if (t < TB) kms[t] = (k0 + t < S && (!kmask || kmask[b * Sp + k0 + t] != 0)) ? 1 : 0;
    __syncthreads();
    const int keyw = k0 + 16 * w;            // this wave's first key
    const bool has_keys = keyw < S;          // wave-uniform
    const int key = keyw + i16;
    bf16x8 fk[2], fv[2];                     // B operands (n = key), loop-invariant over the query blocks
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      fk[kk] = frag_rows64(Ks, 16 * w, kk, lane);
      fv[kk] = frag_rows64(Vs, 16 * w, kk, lane);
    }
    const bool kvalid = key < S && kms[16 * w + i16];
    f32x4 dvT[4], dkT[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dvT[dt] = dkT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // runtime loop (one copy of the body); only the dQ accumulators are indexed statically, at the end of the pair
#pragma unroll 1
    for (int qb = causal ? kb : 0; qb < nblk; ++qb) {
      {
        const int q0 = qb * TB;
        if (kb != 0 || qb != 0) {   // (pair (0, 0) was staged with the key block)
          stage_rows(Qs, base, ld, q0, S, t);
          stage_rows(dOs, dobase, H, q0, S, t);
          __syncthreads();
        }
        // ---- phase A: S, dP (key on the lane) -> P, dS -> dV^t, dK^t; dS^t to LDS
        const int nqt = min(8, (S - q0 + 15) >> 4);          // 16-query tiles with a real row
        // on the diagonal block queries below this wave's first key see none of its keys
        const int mi_lo = (causal && qb == kb) ? w : 0;
        u32x2 dsp[ALIAS ? 8 : 1];
        auto phase_a = [&](const int s) {
            float pd[2][4], ds[2][4];
            if (2 * s < nqt && 2 * s + 1 >= mi_lo) {
              f32x4 sa[2], da[2];
#pragma unroll
              for (int tt = 0; tt < 2; ++tt) {
                sa[tt] = da[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                  sa[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows64(Qs, (2 * s + tt) * 16, kk, lane), fk[kk],
                                                                   sa[tt], 0, 0, 0);
                  da[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows64(dOs, (2 * s + tt) * 16, kk, lane), fv[kk],
                                                                   da[tt], 0, 0, 0);
                }
              }
#pragma unroll
              for (int tt = 0; tt < 2; ++tt) {
                const int ql = (2 * s + tt) * 16 + g * 4;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lses + q0 + ql);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dels + q0 + ql);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int q = q0 + ql + r;
                  const bool ok = kvalid && q < S && (!causal || key <= q);
                  const float pu = ok ? __expf(sa[tt][r] * scale - l4[r]) : 0.f;  // undropped probability
                  // (the key runs along the lanes here, so each element has its own pair hash)
                  const float m = drop.on() ? drop.mul((((unsigned)b * heads + h) * Sp + q) * Sp + key) : 1.f;
                  ds[tt][r] = pu * (da[tt][r] * m - d4[r]) * scale;               // dP = dP_dropped * m
                  pd[tt][r] = pu * m;                                              // dV uses the dropped probabilities
                }
              }
              const bf16x8 fp = pack_acc(pd[0], pd[1]);
              const bf16x8 fds = pack_acc(ds[0], ds[1]);
#pragma unroll
              for (int dt = 0; dt < 4; ++dt) {
                dvT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_acc(dOs, QS, s * 32, dt * 16, lane), fp, dvT[dt],
                                                                  0, 0, 0);
                dkT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr_acc(Qs, QS, s * 32, dt * 16, lane), fds, dkT[dt],
                                                                  0, 0, 0);
              }
            } else {
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[tt][r] = 0.f;
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
              u32x2 pk;
              pk[0] = pack2(ds[tt][0], ds[tt][1]);
              pk[1] = pack2(ds[tt][2], ds[tt][3]);
              if constexpr (ALIAS) dsp[2 * s + tt] = pk;   // V / Q are still being read: held until the barrier
              else *reinterpret_cast<u32x2*>(mine + i16 * PS + ((2 * s + tt) * 16 + g * 4) * 2) = pk;
            }
        };
        if (has_keys) {
          if constexpr (ALIAS) {
#pragma unroll
            for (int s = 0; s < 4; ++s) phase_a(s);
          } else {
#pragma unroll 1
            for (int s = 0; s < 4; ++s) phase_a(s);
          }
        } else {  // no real key in this wave's rows: the dQ product must read zeros there
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) {
            if constexpr (ALIAS) dsp[mi] = (u32x2){0u, 0u};
            else *reinterpret_cast<u32x2*>(mine + i16 * PS + (mi * 16 + g * 4) * 2) = (u32x2){0u, 0u};
          }
        }
        if constexpr (ALIAS) {
          __syncthreads();  // every wave is done with V and Q: their space takes the dS^t image
#pragma unroll
          for (int mi = 0; mi < 8; ++mi)
            *reinterpret_cast<u32x2*>(mine + i16 * PS + (mi * 16 + g * 4) * 2) = dsp[mi];
        }
        __syncthreads();
        // ---- phase B: dQ[q][d] += dS[q][key] K[key][d] for this wave's 16 queries of the block
        if (q0 + 16 * w < S) {
          int nks = (min(TB, S - k0) + 31) >> 5;
          if (causal && qb == kb) nks = min(nks, (w >> 1) + 1);
          f32x4 dqt[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) dqt[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
          for (int ks = 0; ks < nks; ++ks) {
            const bf16x8 fa = frag_tr(dSs, PS, ks * 32, 16 * w, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              dqt[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, frag_tr(Ks, QS, ks * 32, nt * 16, lane), dqt[nt], 0, 0,
                                                                0);
          }
#pragma unroll
          for (int j = 0; j < NQB; ++j)
            if (j == qb) {  // wave-uniform select keeps dq[][] in registers (no runtime-indexed vector array)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) dq[j][nt] += dqt[nt];
            }
        }
        __syncthreads();  // Qs / dOs / dSs are re-filled by the next pair
      }
    }
    // ---- dK, dV of this key block: dkT[dt][r] = dK[key = keyw + i16][d = dt*16 + 4g + r]
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      u32x2 pk, pv;
      pk[0] = pack2(dkT[dt][0], dkT[dt][1]);
      pk[1] = pack2(dkT[dt][2], dkT[dt][3]);
      pv[0] = pack2(dvT[dt][0], dvT[dt][1]);
      pv[1] = pack2(dvT[dt][2], dvT[dt][3]);
      *reinterpret_cast<u32x2*>(mine + i16 * PS + (dt * 16 + g * 4) * 2) = pk;
      *reinterpret_cast<u32x2*>(mine + i16 * PS + 128 + (dt * 16 + g * 4) * 2) = pv;
    }
    if (has_keys) {
      const int nv = min(16, S - keyw);
      store_rows16(mine, PS, dbase_g + H, ld, keyw, nv, lane);
      store_rows16(mine + 128, PS, dbase_g + 2 * H, ld, keyw, nv, lane);
    }
  }

  // ---- dQ of every query block: dq[qb][nt][r] = dQ[q = qb*128 + 16w + 4g + r][d = nt*16 + i16]
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int qw = qb * TB + 16 * w;
    if (qw < S) {  // wave-uniform
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<bf16_t*>(mine + (g * 4 + r) * PS + (nt * 16 + i16) * 2) = f2bf(dq[qb][nt][r]);
      store_rows16(mine, PS, dbase_g, ld, qw, min(16, S - qw), lane);
    }
  }
}

constexpr size_t TFWD_LDS = 2 * TILE_QKV + TB;
constexpr size_t tbwd_lds(int nqb) { return 4 * TILE_QKV + TILE_P + 2 * (size_t)nqb * TB * sizeof(float) + TB; }

constexpr size_t TBWD_ALIAS_LDS = 4 * TILE_QKV + 2 * (size_t)TB * sizeof(float) + TB;

// One-block sequences (S <= 128: every training shape) take the aliased layout, two workgroups per CU.
template <int NQB>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_mask, int B,
               int S, int heads, int causal, void* dqkv, Drop drop, const int32_t* cu, hipStream_t s) {
  constexpr bool ALIAS = NQB == 1;
  constexpr size_t lds = ALIAS ? TBWD_ALIAS_LDS : tbwd_lds(NQB);
  static const hipError_t attr = hipFuncSetAttribute((const void*)attn_bwd_tiled_kernel<NQB, ALIAS>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) {
    set_error("attention (tiled backward): cannot raise dynamic LDS limit");
    return PGCA_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((attn_bwd_tiled_kernel<NQB, ALIAS>), dim3(heads, B), dim3(TNT), lds, s, (const bf16_t*)qkv,
                     (const bf16_t*)out, (const bf16_t*)dout, lse, key_mask, S, heads, causal, (bf16_t*)dqkv, drop, cu, S);
  return check_launch("pgca_attention_bwd(tiled)");
}

}  // namespace

namespace pgca {

int attention_fwd_tiled(const void* qkv, const int32_t* key_mask, int B, int S, int heads, int causal, void* out,
                        float* lse, uint32_t drop_seed, uint32_t drop_threshold, float drop_scale, const int32_t* cu,
                        void* stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)attn_fwd_tiled_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)TFWD_LDS);
  if (attr != hipSuccess) {
    set_error("attention (tiled forward): cannot raise dynamic LDS limit");
    return PGCA_ERR_LAUNCH;
  }
  if (S <= TB) {  // one key tile: the four-wave kernel
    hipLaunchKernelGGL(attn_fwd_one_kernel, dim3(heads, B), dim3(FNT), TFWD_LDS, (hipStream_t)stream, (const bf16_t*)qkv,
                       key_mask, S, heads, causal, (bf16_t*)out, lse, Drop{drop_seed, drop_threshold, drop_scale}, cu, S);
    return check_launch("pgca_attention_fwd(one tile)");
  }
  const int nqb = (S + TB - 1) / TB;
  hipLaunchKernelGGL(attn_fwd_tiled_kernel, dim3(heads, B, nqb), dim3(TNT), TFWD_LDS, (hipStream_t)stream,
                     (const bf16_t*)qkv, key_mask, S, heads, causal, (bf16_t*)out, lse,
                     Drop{drop_seed, drop_threshold, drop_scale}, cu, S);
  return check_launch("pgca_attention_fwd(tiled)");
}

int attention_bwd_tiled(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* key_mask,
                        int B, int S, int heads, int causal, void* dqkv, uint32_t drop_seed, uint32_t drop_threshold,
                        float drop_scale, const int32_t* cu, void* stream) {
  const Drop d{drop_seed, drop_threshold, drop_scale};
  hipStream_t s = (hipStream_t)stream;
  const int nqb = (S + TB - 1) / TB;
  switch (nqb) {
    case 1: return launch_bwd<1>(qkv, out, dout, lse, key_mask, B, S, heads, causal, dqkv, d, cu, s);
    case 2: return launch_bwd<2>(qkv, out, dout, lse, key_mask, B, S, heads, causal, dqkv, d, cu, s);
    case 3: return launch_bwd<3>(qkv, out, dout, lse, key_mask, B, S, heads, causal, dqkv, d, cu, s);
    case 4: return launch_bwd<4>(qkv, out, dout, lse, key_mask, B, S, heads, causal, dqkv, d, cu, s);
    default:
      set_error("pgca_attention_bwd: S=%d exceeds the %d-token limit of the register-resident dQ accumulators", S,
                PGCA_ATTN_MAX_S);
      return PGCA_ERR_INVALID;
  }
}

}  // namespace pgca
