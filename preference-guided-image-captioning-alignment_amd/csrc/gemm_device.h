// Device-side building blocks shared by the GEMM translation units (gemm.hip, gemm_phase.hip): operand staging
// geometry, MFMA fragment readers, the fused epilogues and the LDS-DMA issue helpers.  Everything here has
// internal linkage; the cross-TU symbols are pgca::launch_gemm256s (gemm_phase.hip) and pgca::gemm_tuning (gemm.hip).
#pragma once
#include <stdlib.h>

#include "common.h"

namespace pgca {
// 256 x 256, 8 waves, phase-staggered wave groups, 4-stage BK=32 ring (gemm_phase.hip).
int launch_gemm256s(const pgca_gemm_args& a, int ntm, int ntn, int nk_per_split, int nsplit, void* stream);

// Process-wide dispatch knobs (pgca_set_option / environment, read ONCE): nothing on the launch path calls getenv.
struct GemmTuning {
  int tile;      // 0 = automatic, 128 / 256 = force that kernel family      (PGCA_GEMM_TILE)
  int schedule;  // -1 = automatic, 0 = 2-stage BK=64 loop, 6 = phase-staggered (PGCA_GEMM_RING)
  int group;     // 1 = the four weight gradients of a block in one grid        (PGCA_GEMM_NO_GROUP inverts)
  int stagger;   // start delay step of the first wave of workgroups, in units of 1024 clocks (PGCA_GEMM_STAGGER)
};
GemmTuning& gemm_tuning();
}  // namespace pgca

using namespace pgca;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffff0, 0x00020000);
}

// ---- per-thread staging geometry of one operand tile -----------------------------------------
// KS = 0: operand is [rows, K] (K contiguous); tile image [128 rows][64 k], 128-B rows,
//         16-B chunk c of row r stored at chunk (c ^ (r & 7)).
// KS = 1: operand is [K, cols] (K strided);   tile image [64 k][128 cols], 256-B rows,
//         32-B slot s of row k stored at slot (s ^ h(k)), h(k) = (k&3) | ((k>>3)&1)<<2.
template <int KS>
struct Stage {
  unsigned goff[4];   // byte offset of this thread's 4 chunks relative to the tile base pointer
  unsigned loff[4];   // byte offset in the LDS image
  bool vspace[4];     // row (KS=0) / column (KS=1) inside the matrix
  int kpos[4];        // k offset inside the tile of each chunk (for the K-edge test)

  __device__ __forceinline__ void init(int t, int ld, int origin, int extent) {
    if (KS == 0) {
      const int r = t >> 3, c = t & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r + 32 * i;
        goff[i] = (unsigned)(row * ld + c * 8) * 2u;
        loff[i] = (unsigned)(row * 128 + ((c ^ (row & 7)) << 4));
        vspace[i] = (origin + row) < extent;
        kpos[i] = c * 8;
      }
    } else {
      const int kr = t >> 4, c16 = t & 15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kr + 16 * i;
        const int h = (k & 3) | (((k >> 3) & 1) << 2);
        goff[i] = (unsigned)(k * ld + c16 * 8) * 2u;
        loff[i] = (unsigned)(k * 256 + (((c16 >> 1) ^ h) << 5) + ((c16 & 1) << 4));
        vspace[i] = (origin + c16 * 8) < extent;
        kpos[i] = k;
      }
    }
  }
  __device__ __forceinline__ void load(const bf16_t* base, int krem, u32x4 (&r)[4]) const {
    __amdgpu_buffer_rsrc_t rs = make_rsrc(base);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = vspace[i] && (kpos[i] < krem);
      r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? goff[i] : OOB, 0, 0);
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds, const u32x4 (&r)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(lds + loff[i]) = r[i];
  }
};

// ---- fragment readers ---------------------------------------------------------------------------
// Returns the MFMA 16x16x32 operand fragment of 16-row (or 16-col) sub-tile `sub` of the wave's
// 64-wide strip starting at `wbase`, k-step kk (0,1) of the 64-deep tile.
template <int KS, int KSTRIDE = 256>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* lds, int wbase, int sub, int kk, int lane) {
  if (KS == 0) {
    const int row = wbase + sub * 16 + (lane & 15);
    const int c = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((c ^ (lane & 7)) << 4));
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int k = kk * 32 + 8 * g + q;
    const int h = q | ((g & 1) << 2);
    const int s32 = (wbase >> 4) + sub;
    const unsigned char* a0 = lds + k * KSTRIDE + ((s32 ^ h) << 5) + 8 * p;
    return tr_frag(a0, a0 + 4 * KSTRIDE);
  }
}

// ---- epilogues ------------------------------------------------------------------------------------
// Accumulator element acc[mi][ni][r] of wave (wm, wn) is C[m0 + wm*64 + mi*16 + (lane>>4)*4 + r]
//                                                        [n0 + wn*64 + ni*16 + (lane&15)].
// max / sum over the 16 lanes of a DPP row (the 16 columns a 16x16 MFMA tile gives one output row), every lane gets
// the result.  quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: four VALU ops, no LDS crossbar
// (`__shfl_xor` compiles to ds_bpermute: ~10x the latency, and the LM-head epilogue does 32 reductions per block).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}

__device__ __forceinline__ void epilogue_rowstats(const pgca_gemm_args& a, f32x4 (&acc)[4][4], int m0, int n0, int tn,
                                                  int wm, int wn, int lane) {
  const int rbase = m0 + wm * 64 + (lane >> 4) * 4;
  const int cbase = n0 + wn * 64 + (lane & 15);
  const int part = (n0 >> 6) + wn;  // one partial per 64-column strip
  // the 16 target ids of this lane's rows, requested before anything depends on them (one round trip, not 16)
  long long tg[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + mi * 16 + r;
      tg[mi][r] = (row < a.M && a.targets) ? a.targets[row] : -1;
    }
  float bs[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) bs[ni] = (a.bias && cbase + ni * 16 < a.N) ? a.bias[cbase + ni * 16] : 0.f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + mi * 16 + r;
      const long long tgt = tg[mi][r];
      float x[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = cbase + ni * 16;
        const float t = a.alpha * acc[mi][ni][r] + bs[ni];
        if (col == tgt && a.target_val) a.target_val[row] = t;
        x[ni] = col < a.N ? t : -INFINITY;
      }
      const float mx = row16_max(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
      float sm = 0.f;
      if (mx > -INFINITY) sm = __expf(x[0] - mx) + __expf(x[1] - mx) + __expf(x[2] - mx) + __expf(x[3] - mx);
      sm = row16_sum(sm);
      if ((lane & 15) == 0 && row < a.M) {
        a.stat_max[(size_t)row * a.stat_ld + part] = mx;
        a.stat_sum[(size_t)row * a.stat_ld + part] = sm;
      }
    }
  }
}

__device__ __forceinline__ bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Column sums of one 64 x 64 block (pgca_gemm_args::colsum_part): every lane holds the sums of its 8 columns over the
// rows it handled; the eight lanes that share those columns (lane bits 3..5 = row within the slab) are folded with three
// DPP-free shuffles and lane-row 0 writes row `brow` of the partial matrix.
__device__ __forceinline__ void colsum_flush(const pgca_gemm_args& a, float (&cs)[8], int brow, int col, int ncols,
                                             int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cs[j] += __shfl_xor(cs[j], 8);
    cs[j] += __shfl_xor(cs[j], 16);
    cs[j] += __shfl_xor(cs[j], 32);
  }
  if ((lane >> 3) == 0 && brow * 64 < a.M) {
    float* p = a.colsum_part + (size_t)brow * a.ld_colsum + col;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (col + j < ncols) p[j] = cs[j];
  }
}

// Everything that happens to 8 consecutive output columns of one row.  `nv` = number of valid columns (1..8);
// the 16-byte vector paths are taken when nv == 8 and the row start is 16-B aligned, else element-wise.
template <int EPI>
__device__ __forceinline__ void finish8(const pgca_gemm_args& a, int row, int col, float (&v)[8], int nv, float lse,
                                        float rscale, long long tgt) {
  const bool full = nv == 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] *= a.alpha;
  if (a.bias) {
    const float* bp = a.bias + col;
    if (full && al16(bp)) {
      const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < nv && col + j < a.N) v[j] += bp[j];
    }
  }
  if (EPI == PGCA_EPI_GELU_NEW_D) {
    float dy[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gelu_new_both(v[j], v[j], dy[j]);
    if (a.aux_out) {
      bf16_t* p = reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < nv) p[j] = f2bf(dy[j]);
    }
  } else if (EPI == PGCA_EPI_GELU_NEW || EPI == PGCA_EPI_QUICK_GELU) {
    if (a.aux_out) {
      bf16_t* p = reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col;
      if (full && al16(p)) {
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = f2bf(v[j]);
        *reinterpret_cast<bf16x8*>(p) = t;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < nv) p[j] = f2bf(v[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = EPI == PGCA_EPI_GELU_NEW ? gelu_new(v[j]) : quick_gelu(v[j]);
  } else if (EPI == PGCA_EPI_RELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (EPI == PGCA_EPI_TANH) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fast_tanh(v[j]);
    if (a.aux_out) {  // undropped tanh, operand of DTANH in the backward
      bf16_t* p = reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < nv) p[j] = f2bf(v[j]);
    }
  } else if (EPI == PGCA_EPI_DGELU_NEW || EPI == PGCA_EPI_DRELU || EPI == PGCA_EPI_DTANH || EPI == PGCA_EPI_DQUICK_GELU ||
             EPI == PGCA_EPI_MUL_AUX) {
    const bf16_t* p = reinterpret_cast<const bf16_t*>(a.aux_in) + (size_t)row * a.ld_aux + col;
    float x[8];
    if (full && al16(p)) {
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = bf2f(t[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = j < nv ? bf2f(p[j]) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (EPI == PGCA_EPI_DGELU_NEW) v[j] *= dgelu_new(x[j]);
      else if (EPI == PGCA_EPI_MUL_AUX) v[j] *= x[j];
      else if (EPI == PGCA_EPI_DQUICK_GELU) v[j] *= dquick_gelu(x[j]);
      else if (EPI == PGCA_EPI_DRELU) v[j] = x[j] > 0.f ? v[j] : 0.f;
      else v[j] *= 1.f - x[j] * x[j];
    }
  } else if (EPI == PGCA_EPI_DLOGITS) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = (col + j) < a.N ? rscale * (__expf(v[j] - lse) - ((col + j) == tgt ? 1.f : 0.f)) : 0.f;
  }
  if (a.drop_threshold) {
    const Drop d{a.drop_seed, a.drop_threshold, a.drop_scale};
    const unsigned drow = a.drop_rows ? (unsigned)a.drop_rows[row] : (unsigned)row;  // packed rows: hash of the padded position
    const unsigned base = drow * (unsigned)a.N + (unsigned)col;
    float m0[4], m1[4];
    d.mul4(base, m0);
    d.mul4(base + 4u, m1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] *= m0[j]; v[j + 4] *= m1[j]; }
  }
  if (a.residual) {
    const float* p = a.residual + (size_t)row * a.ld_res + col;
    if (full && al16(p)) {
      const float4 b0 = *reinterpret_cast<const float4*>(p), b1 = *reinterpret_cast<const float4*>(p + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < nv) v[j] += p[j];
    }
  }
  if (a.out_f32) {
    float* p = a.out_f32 + (size_t)row * a.ld_out_f32 + col;
    if (full && al16(p)) {
      float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
      if (a.accumulate) {
        const float4 c0 = *reinterpret_cast<const float4*>(p), c1 = *reinterpret_cast<const float4*>(p + 4);
        o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w; o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
      }
      *reinterpret_cast<float4*>(p) = o0;
      *reinterpret_cast<float4*>(p + 4) = o1;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < nv) p[j] = a.accumulate ? p[j] + v[j] : v[j];
    }
  }
  if (a.out_bf16) {
    bf16_t* p = reinterpret_cast<bf16_t*>(a.out_bf16) + (size_t)row * a.ld_out_bf16 + col;
    if (full && al16(p)) {
      bf16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = f2bf(v[j]);
      *reinterpret_cast<bf16x8*>(p) = t;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < nv) p[j] = f2bf(v[j]);
    }
  }
  if (EPI == PGCA_EPI_DLOGITS && a.aux_out) {  // low half of the hi/lo bf16 split of the result (NT-Xent: f32-grade G)
    bf16_t* p = reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j < nv) p[j] = f2bf(v[j] - bf2f(f2bf(v[j])));
  }
}

// ---- wave-private staging slab -------------------------------------------------------------------------
// The accumulator block crosses LDS one 16-row MFMA tile row at a time: a slab of 16 x 64 f32 = 4 KiB per wave (32 KiB
// for the eight waves - ONE stage of the 256^2 operand ring, so a persistent kernel can keep the other three stages filled
// with the next tile's operands while it stores).  No padding: float4 group q of row r sits at group q ^ (r & 7), which makes
// the b32 writes of the MFMA layout (two rows x 16 columns per 32 lanes: bit 4 of the column differs) and the b128 row reads
// (the 16-lane service groups touch 16 distinct groups) conflict-free.
constexpr int CB_LD = 64, CB_ROWS = 16;
constexpr int CB_WAVE_FLOATS = CB_LD * CB_ROWS;  // 1024 floats = 4 KiB
__device__ __forceinline__ int cb_idx(int row, int col) { return row * CB_LD + (col ^ ((row & 7) << 2)); }
// slab <- tile row mi of the block (rows mi*16 .. mi*16+15)
__device__ __forceinline__ void cb_write(float* cbuf, const f32x4 (&acc)[4][4], int mi, int lane) {
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) cbuf[cb_idx((lane >> 4) * 4 + r, ni * 16 + (lane & 15))] = acc[mi][ni][r];
}
// 8 consecutive columns (lane & 7) * 8 .. +7 of slab row lr
__device__ __forceinline__ void cb_read8(const float* cbuf, int lr, int lane, float (&v)[8]) {
  const int q = (lane & 7) * 2, x = lr & 7;
  const float4 c0 = *reinterpret_cast<const float4*>(cbuf + lr * CB_LD + ((q ^ x) << 2));
  const float4 c1 = *reinterpret_cast<const float4*>(cbuf + lr * CB_LD + (((q + 1) ^ x) << 2));
  v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
}

// ---- fast epilogue ---------------------------------------------------------------------------------
// Interior, 16-B-aligned 64 x 64 blocks (all of the hot path).  What the general path below costs is not
// arithmetic but LATENCY: it loads the bias / residual stream / saved activation of 8 rows, waits, finishes them,
// stores, and only then loads the next 8 rows - 16 dependent HBM round trips per 64 x 64 block pair, with the
// matrix pipe idle (s_memtime: 35-66k clocks of epilogue against 52k of main loop at K = 1024).  Here every
// global operand of the whole block is requested up front (one round trip), no lane predicates, no block barriers
// (the staging buffer is private to the wave and a wave's LDS operations execute in order).
enum { PF_NONE = 0, PF_RES = 1, PF_ACC = 2 };

template <int EPI, int PF>
__device__ __forceinline__ void epilogue_store_fast(const pgca_gemm_args& a, f32x4 (&acc)[4][4], unsigned char* smem,
                                                    int mb, int cb, int lane, int wave) {
  constexpr bool AUXIN = EPI == PGCA_EPI_DGELU_NEW || EPI == PGCA_EPI_DRELU || EPI == PGCA_EPI_DTANH ||
                         EPI == PGCA_EPI_DQUICK_GELU || EPI == PGCA_EPI_MUL_AUX;
  float* cbuf = reinterpret_cast<float*>(smem) + wave * CB_WAVE_FLOATS;
  const int r8 = lane >> 3, cg = (lane & 7) * 8;
  const int col = cb + cg;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (a.bias) {
    b0 = *reinterpret_cast<const float4*>(a.bias + col);
    b1 = *reinterpret_cast<const float4*>(a.bias + col + 4);
  }
  constexpr bool CSUM = EPI == PGCA_EPI_DGELU_NEW || EPI == PGCA_EPI_MUL_AUX;
  float cs[8];
  if constexpr (CSUM) {
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  }
  float4 pre[PF != PF_NONE ? 8 : 1][2];
  bf16x8 ax[AUXIN ? 8 : 1];
  float lse[EPI == PGCA_EPI_DLOGITS ? 8 : 1], rsc[EPI == PGCA_EPI_DLOGITS ? 8 : 1];
  long long tgt[EPI == PGCA_EPI_DLOGITS ? 8 : 1];
  unsigned drow[8];  // row index the dropout hash is keyed on (pgca_gemm_args::drop_rows; dead code without dropout)
#pragma unroll
  for (int i8 = 0; i8 < 8; ++i8) {
    const int row = mb + i8 * 8 + r8;
    drow[i8] = (a.drop_threshold && a.drop_rows) ? (unsigned)a.drop_rows[row] : (unsigned)row;
    if (PF == PF_RES) {
      const float* p = a.residual + (size_t)row * a.ld_res + col;
      pre[PF != PF_NONE ? i8 : 0][0] = *reinterpret_cast<const float4*>(p);
      pre[PF != PF_NONE ? i8 : 0][1] = *reinterpret_cast<const float4*>(p + 4);
    } else if (PF == PF_ACC) {
      const float* p = a.out_f32 + (size_t)row * a.ld_out_f32 + col;
      pre[PF != PF_NONE ? i8 : 0][0] = *reinterpret_cast<const float4*>(p);
      pre[PF != PF_NONE ? i8 : 0][1] = *reinterpret_cast<const float4*>(p + 4);
    }
    if (AUXIN)
      ax[AUXIN ? i8 : 0] =
          *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(a.aux_in) + (size_t)row * a.ld_aux + col);
    if (EPI == PGCA_EPI_DLOGITS) {
      lse[EPI == PGCA_EPI_DLOGITS ? i8 : 0] = a.row_lse[row];
      rsc[EPI == PGCA_EPI_DLOGITS ? i8 : 0] = a.row_scale[row];
      tgt[EPI == PGCA_EPI_DLOGITS ? i8 : 0] = a.targets[row];
    }
  }
  const Drop d{a.drop_seed, a.drop_threshold, a.drop_scale};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    cb_write(cbuf, acc, mi, lane);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i8 = mi * 2 + it;
      const int lr = it * 8 + r8;
      const int row = mb + mi * 16 + lr;
      float v[8];
      cb_read8(cbuf, lr, lane, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= a.alpha;
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      if (EPI == PGCA_EPI_GELU_NEW_D) {
        bf16x8 tq;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float dy;
          gelu_new_both(v[j], v[j], dy);
          tq[j] = f2bf(dy);
        }
        if (a.aux_out)
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col) = tq;
      } else if (EPI == PGCA_EPI_GELU_NEW || EPI == PGCA_EPI_QUICK_GELU) {
        if (a.aux_out) {
          bf16x8 tq;
#pragma unroll
          for (int j = 0; j < 8; ++j) tq[j] = f2bf(v[j]);
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col) = tq;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = EPI == PGCA_EPI_GELU_NEW ? gelu_new(v[j]) : quick_gelu(v[j]);
      } else if (EPI == PGCA_EPI_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if (EPI == PGCA_EPI_TANH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fast_tanh(v[j]);
        if (a.aux_out) {
          bf16x8 tq;
#pragma unroll
          for (int j = 0; j < 8; ++j) tq[j] = f2bf(v[j]);
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col) = tq;
        }
      } else if (AUXIN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = bf2f(ax[AUXIN ? i8 : 0][j]);
          if (EPI == PGCA_EPI_DGELU_NEW) v[j] *= dgelu_new(x);
          else if (EPI == PGCA_EPI_MUL_AUX) v[j] *= x;
          else if (EPI == PGCA_EPI_DQUICK_GELU) v[j] *= dquick_gelu(x);
          else if (EPI == PGCA_EPI_DRELU) v[j] = x > 0.f ? v[j] : 0.f;
          else v[j] *= 1.f - x * x;
        }
      } else if (EPI == PGCA_EPI_DLOGITS) {
        constexpr int q = EPI == PGCA_EPI_DLOGITS ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = (col + j) < a.N ? rsc[q * i8] * (__expf(v[j] - lse[q * i8]) - ((col + j) == tgt[q * i8] ? 1.f : 0.f)) : 0.f;
      }
      if (a.drop_threshold) {
        const unsigned base = drow[i8] * (unsigned)a.N + (unsigned)col;
        float m0[4], m1[4];
        d.mul4(base, m0);
        d.mul4(base + 4u, m1);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] *= m0[j]; v[j + 4] *= m1[j]; }
      }
      if constexpr (CSUM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += v[j];
      }
      if (PF == PF_RES) {
        const float4 p0 = pre[PF != PF_NONE ? i8 : 0][0], p1 = pre[PF != PF_NONE ? i8 : 0][1];
        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
      }
      if (a.out_f32) {
        float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
        if (PF == PF_ACC) {
          const float4 p0 = pre[PF != PF_NONE ? i8 : 0][0], p1 = pre[PF != PF_NONE ? i8 : 0][1];
          o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w; o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
        }
        float* p = a.out_f32 + (size_t)row * a.ld_out_f32 + col;
        *reinterpret_cast<float4*>(p) = o0;
        *reinterpret_cast<float4*>(p + 4) = o1;
      }
      if (a.out_bf16) {
        bf16x8 tq;
#pragma unroll
        for (int j = 0; j < 8; ++j) tq[j] = f2bf(v[j]);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(a.out_bf16) + (size_t)row * a.ld_out_bf16 + col) = tq;
      }
      if (EPI == PGCA_EPI_DLOGITS && a.aux_out) {
        bf16x8 tq;
#pragma unroll
        for (int j = 0; j < 8; ++j) tq[j] = f2bf(v[j] - bf2f(f2bf(v[j])));
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(a.aux_out) + (size_t)row * a.ld_aux + col) = tq;
      }
    }
  }
  if constexpr (CSUM) {
    if (a.colsum_part) colsum_flush(a, cs, mb >> 6, col, a.N, lane);
  }
}

// Uniform (per wave) test for the fast path: block inside the matrix, every pointer it touches 16-B aligned.
template <int EPI>
__device__ __forceinline__ bool epilogue_fast_ok(const pgca_gemm_args& a, int mb, int cb) {
  const int ncols = EPI == PGCA_EPI_DLOGITS ? a.out_cols : a.N;
  if (a.accumulate == 2 || mb + 64 > a.M || cb + 64 > ncols) return false;
  if (a.residual && a.out_f32 && a.accumulate) return false;
  bool ok = true;
  if (a.bias) ok = ok && al16(a.bias);
  if (a.residual) ok = ok && al16(a.residual) && !(a.ld_res & 3);
  if (a.out_f32) ok = ok && al16(a.out_f32) && !(a.ld_out_f32 & 3);
  if (a.out_bf16) ok = ok && al16(a.out_bf16) && !(a.ld_out_bf16 & 7);
  if (a.aux_out) ok = ok && al16(a.aux_out) && !(a.ld_aux & 7);
  if (a.aux_in) ok = ok && al16(a.aux_in) && !(a.ld_aux & 7);
  return ok;
}

// One 64 x 64 accumulator block (origin m0 + wm*64, n0 + wn*64) through the wave-private slab, 16 rows at a time, so every
// global access of the epilogue (bias, saved activations, residual stream, outputs) is a 16-byte vector op on 128..256
// contiguous bytes per row.  No block barriers anywhere: the slab is private to the wave and a wave's LDS operations
// execute in order (the caller guarantees every wave has left the main loop's LDS before the first epilogue starts).
template <int EPI>
__device__ __forceinline__ void epilogue_store(const pgca_gemm_args& a, f32x4 (&acc)[4][4], unsigned char* smem, int m0,
                                               int n0, int wm, int wn, int lane, int wave) {
  // residual / accumulate prefetch modes exist for the plain epilogue only (where the hot path uses them)
  const bool rmw = a.residual || (a.out_f32 && a.accumulate);
  if ((EPI == PGCA_EPI_NONE || !rmw) && epilogue_fast_ok<EPI>(a, m0 + wm * 64, n0 + wn * 64)) {  // wave-uniform
    const int mb = m0 + wm * 64, cb = n0 + wn * 64;
    if (EPI == PGCA_EPI_NONE && a.residual) epilogue_store_fast<PGCA_EPI_NONE, PF_RES>(a, acc, smem, mb, cb, lane, wave);
    else if (EPI == PGCA_EPI_NONE && rmw) epilogue_store_fast<PGCA_EPI_NONE, PF_ACC>(a, acc, smem, mb, cb, lane, wave);
    else epilogue_store_fast<EPI, PF_NONE>(a, acc, smem, mb, cb, lane, wave);
    return;
  }
  float* cbuf = reinterpret_cast<float*>(smem) + wave * CB_WAVE_FLOATS;
  const int ncols = EPI == PGCA_EPI_DLOGITS ? a.out_cols : a.N;
  constexpr bool CSUM = EPI == PGCA_EPI_DGELU_NEW || EPI == PGCA_EPI_MUL_AUX;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    cb_write(cbuf, acc, mi, lane);
    if (EPI == PGCA_EPI_NONE && a.accumulate == 2) {
      // split-K partial: f32 atomic adds, one 256-byte row segment of the tile per wave-instruction
      // (the access shape at which global_atomic_add_f32 runs at its full memory-side rate)
      const int col = n0 + wn * 64 + lane;
      if (col < a.N) {
        for (int lr = 0; lr < CB_ROWS; ++lr) {
          const int row = m0 + wm * 64 + mi * 16 + lr;
          if (row < a.M) atomicAdd(a.out_f32 + (size_t)row * a.ld_out_f32 + col, a.alpha * cbuf[cb_idx(lr, lane)]);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int lr = it * 8 + (lane >> 3);
        const int row = m0 + wm * 64 + mi * 16 + lr;
        const int col = n0 + wn * 64 + (lane & 7) * 8;
        float v[8];
        cb_read8(cbuf, lr, lane, v);
        if (row < a.M && col < ncols) {
          float lse = 0.f, rscale = 0.f;
          long long tgt = -1;
          if (EPI == PGCA_EPI_DLOGITS) {
            lse = a.row_lse[row];
            rscale = a.row_scale[row];
            tgt = a.targets[row];
          }
          const int nv = ncols - col < 8 ? ncols - col : 8;
          finish8<EPI>(a, row, col, v, nv, lse, rscale, tgt);
          if constexpr (CSUM) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[j] += j < nv ? v[j] : 0.f;
          }
        }
      }
    }
  }
  if constexpr (CSUM) {
    if (a.colsum_part) colsum_flush(a, cs, (m0 + wm * 64) >> 6, n0 + wn * 64 + (lane & 7) * 8, ncols, lane);
  }
}


// Runtime epilogue selection for one 64 x 64 accumulator block whose first row is `mh` (wave column wn).
__device__ __forceinline__ void run_epilogue(const pgca_gemm_args& a, f32x4 (&acc)[4][4], unsigned char* smem, int mh,
                                             int n0, int tn, int wn, int lane, int wave) {
  switch (a.epilogue) {
    case PGCA_EPI_ROWSTATS: epilogue_rowstats(a, acc, mh, n0, tn, 0, wn, lane); break;
    case PGCA_EPI_GELU_NEW: epilogue_store<PGCA_EPI_GELU_NEW>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_QUICK_GELU: epilogue_store<PGCA_EPI_QUICK_GELU>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_RELU: epilogue_store<PGCA_EPI_RELU>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_TANH: epilogue_store<PGCA_EPI_TANH>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_DGELU_NEW: epilogue_store<PGCA_EPI_DGELU_NEW>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_DRELU: epilogue_store<PGCA_EPI_DRELU>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_DTANH: epilogue_store<PGCA_EPI_DTANH>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_DLOGITS: epilogue_store<PGCA_EPI_DLOGITS>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_DQUICK_GELU: epilogue_store<PGCA_EPI_DQUICK_GELU>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_GELU_NEW_D: epilogue_store<PGCA_EPI_GELU_NEW_D>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    case PGCA_EPI_MUL_AUX: epilogue_store<PGCA_EPI_MUL_AUX>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
    default: epilogue_store<PGCA_EPI_NONE>(a, acc, smem, mh, n0, 0, wn, lane, wave); break;
  }
}

constexpr int BM2 = 256, BN2 = 256;
constexpr int TILE2_BYTES = 256 * 64 * 2;

template <int KS>
struct Dma {
  unsigned goff[4];
  __device__ __forceinline__ void init(int lane, int wave, int ld, int origin, int extent) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wave * 4 + i;  // 1-KiB piece of the 32-KiB tile image
      if (KS == 0) {               // [256 rows][64 k]: piece = 8 rows x 128 B
        const int r = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ (lane >> 3);
        const int rg = min(origin + r, extent - 1) - origin;
        goff[i] = (unsigned)(rg * ld + c * 8) * 2u;
      } else {                     // [64 k][256 cols]: piece = 2 k-rows x 512 B
        const int k = 2 * j + (lane >> 5);
        const int c16 = lane & 31;
        const int h = (k & 3) | (((k >> 3) & 1) << 2);
        const int col = (((c16 >> 1) ^ h) << 4) + ((c16 & 1) << 3);
        const int cg = min(origin + col, extent - 8) - origin;
        goff[i] = (unsigned)(k * ld + cg) * 2u;
      }
    }
  }
  // The DMA is issued from inline asm on purpose: hipcc tracks a builtin LDS-DMA as a pending LDS write and
  // drains it (s_waitcnt vmcnt(0)) in front of the next ds_read, which would serialise the copy of tile t+1 with
  // the MFMAs of tile t.  Untracked, it stays in flight across the whole compute phase; dma_wait() retires it
  // right before the barrier that hands the buffer to the readers.  (M0 = LDS byte address of lane 0's 16 bytes.)
  __device__ __forceinline__ void issue(const bf16_t* base, unsigned char* tile, int wave) const {
    const unsigned long long b = (unsigned long long)base;
    u32x4 rs;
    rs[0] = (unsigned)b;
    rs[1] = (unsigned)(b >> 32) & 0xffffu;
    rs[2] = 0x7ffffff0u;
    rs[3] = 0x00020000u;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(tile) + (unsigned)wave * 4096u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :
                   : "s"(lds0 + i * 1024u), "v"(goff[i]), "s"(rs)
                   : "memory");
    }
  }
};

__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int LA>
__device__ __forceinline__ void mma_half(const unsigned char* la, int row0, int kk, int lane, const bf16x8 (&fb)[4],
                                         f32x4 (&acc)[4][4]) {
  bf16x8 fa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = read_frag<LA, 512>(la, row0, i, kk, lane);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
}


}  // namespace
