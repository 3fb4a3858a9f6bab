// bf16 MFMA GEMM for gfx950 (MI355X): C[M,N] = epilogue(alpha * A·B + bias).
//
// One kernel template serves the three operand layouts of the hot path (pgca_hip.h):
//   NT  A[M,K] · B[N,K]^t   both operands contiguous along K   -> ds_read_b128 fragments
//   NN  A[M,K] · B[K,N]     B strided along K                  -> ds_read_b64_tr_b16 fragments
//   TN  A[K,M]^t · B[K,N]   both strided along K (wgrad X^t dY) -> ds_read_b64_tr_b16 fragments
// so forward, dgrad and wgrad all read the reference's native weight layouts
// (Conv1D [in,out], nn.Linear [out,in]) from ONE bf16 mirror - no transposed copies.
//
// Geometry: 128x128 block tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 =
// 4x4 MFMA 16x16x32 accumulators.  Global -> VGPR (buffer loads, OOB lanes read 0) -> LDS
// (XOR-swizzled, conflict-free for both fragment read kinds), double-buffered: the next
// tile's global loads are issued before the current tile's MFMAs and written to the other
// LDS buffer after them (one barrier per K tile).
#include "common.h"

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
template <int KS>
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
    const unsigned char* a0 = lds + k * 256 + ((s32 ^ h) << 5) + 8 * p;
    return tr_frag(a0, a0 + 1024);
  }
}

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const pgca_gemm_args a, int ntm, int ntn) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TILE_BYTES];

  // XCD-aware tile order: blocks that share an XCD (bid % 8) walk a contiguous run of tiles,
  // M fastest, so an XCD's resident blocks share B panels in its private L2.
  const int nwg = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid % ntm, tn = bid / ntm;
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B);

  Stage<LA> sa;
  Stage<LB> sb;
  sa.init(t, a.lda, m0, LA == 0 ? a.M : ((a.M + 7) & ~7));
  sb.init(t, a.ldb, n0, LB == 0 ? a.N : ((a.N + 7) & ~7));
  const bf16_t* abase = LA == 0 ? A + (size_t)m0 * a.lda : A + m0;
  const bf16_t* bbase = LB == 0 ? B + (size_t)n0 * a.ldb : B + n0;
  const size_t astep = LA == 0 ? (size_t)BK : (size_t)BK * a.lda;
  const size_t bstep = LB == 0 ? (size_t)BK : (size_t)BK * a.ldb;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (a.K + BK - 1) / BK;
  u32x4 ra[4], rb[4];
  sa.load(abase, a.K, ra);
  sb.load(bbase, a.K, rb);
  sa.store(smem[0][0], ra);
  sb.store(smem[0][1], rb);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) {
      sa.load(abase + (size_t)(kt + 1) * astep, a.K - (kt + 1) * BK, ra);
      sb.load(bbase + (size_t)(kt + 1) * bstep, a.K - (kt + 1) * BK, rb);
    }
    const unsigned char* la = smem[cur][0];
    const unsigned char* lb = smem[cur][1];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<LA>(la, wm * 64, i, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<LB>(lb, wn * 64, j, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      sa.store(smem[cur ^ 1][0], ra);
      sb.store(smem[cur ^ 1][1], rb);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------------------- epilogue
  const int rbase = m0 + wm * 64 + (lane >> 4) * 4;  // + mi*16 + r
  const int cbase = n0 + wn * 64 + (lane & 15);      // + ni*16
  const int epi = a.epilogue;

  if (epi == PGCA_EPI_ROWSTATS) {
    const int part = tn * 2 + wn;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + mi * 16 + r;
        const long long tgt = (row < a.M && a.targets) ? a.targets[row] : -1;
        float v[4];
        float mx = -INFINITY;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int col = cbase + ni * 16;
          float x = a.alpha * acc[mi][ni][r];
          if (a.bias && col < a.N) x += a.bias[col];
          v[ni] = col < a.N ? x : -INFINITY;
          mx = fmaxf(mx, v[ni]);
          if (col == tgt && a.target_val) a.target_val[row] = x;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sm = 0.f;
        if (mx > -INFINITY) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) sm += __expf(v[ni] - mx);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o);
        if ((lane & 15) == 0 && row < a.M) {
          a.stat_max[(size_t)row * a.stat_ld + part] = mx;
          a.stat_sum[(size_t)row * a.stat_ld + part] = sm;
        }
      }
    }
    return;
  }

  bf16_t* ob = reinterpret_cast<bf16_t*>(a.out_bf16);
  const bf16_t* auxi = reinterpret_cast<const bf16_t*>(a.aux_in);
  bf16_t* auxo = reinterpret_cast<bf16_t*>(a.aux_out);
  const int ncol_store = (epi == PGCA_EPI_DLOGITS) ? a.out_cols : a.N;

#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + mi * 16 + r;
      if (row >= a.M) continue;
      float lse = 0.f, rscale = 0.f;
      long long tgt = -1;
      if (epi == PGCA_EPI_DLOGITS) {
        lse = a.row_lse[row];
        rscale = a.row_scale[row];
        tgt = a.targets[row];
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = cbase + ni * 16;
        if (col >= ncol_store) continue;
        float v = a.alpha * acc[mi][ni][r];
        if (a.bias && col < a.N) v += a.bias[col];
        switch (epi) {
          case PGCA_EPI_GELU_NEW:
            if (auxo) auxo[(size_t)row * a.ld_aux + col] = f2bf(v);
            v = gelu_new(v);
            break;
          case PGCA_EPI_QUICK_GELU:
            if (auxo) auxo[(size_t)row * a.ld_aux + col] = f2bf(v);
            v = quick_gelu(v);
            break;
          case PGCA_EPI_RELU: v = fmaxf(v, 0.f); break;
          case PGCA_EPI_TANH: v = fast_tanh(v); break;
          case PGCA_EPI_DGELU_NEW: v *= dgelu_new(bf2f(auxi[(size_t)row * a.ld_aux + col])); break;
          case PGCA_EPI_DRELU: v = bf2f(auxi[(size_t)row * a.ld_aux + col]) > 0.f ? v : 0.f; break;
          case PGCA_EPI_DTANH: {
            const float y = bf2f(auxi[(size_t)row * a.ld_aux + col]);
            v *= 1.f - y * y;
          } break;
          case PGCA_EPI_DLOGITS:
            v = col < a.N ? rscale * (__expf(v - lse) - (col == tgt ? 1.f : 0.f)) : 0.f;
            break;
          default: break;
        }
        if (a.residual) v += a.residual[(size_t)row * a.ld_res + col];
        if (a.out_f32) {
          float* p = a.out_f32 + (size_t)row * a.ld_out_f32 + col;
          *p = a.accumulate ? (*p + v) : v;
        }
        if (ob) ob[(size_t)row * a.ld_out_bf16 + col] = f2bf(v);
      }
    }
  }
}

__global__ void rowstats_combine_kernel(const float* __restrict__ smax, const float* __restrict__ ssum, int stat_ld,
                                        int nparts, const float* __restrict__ tval, int M, float* __restrict__ lse,
                                        float* __restrict__ out_lp) {
  // one wave per row
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float mx = -INFINITY;
  for (int i = lane; i < nparts; i += 64) mx = fmaxf(mx, smax[(size_t)row * stat_ld + i]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < nparts; i += 64) {
    const float pm = smax[(size_t)row * stat_ld + i];
    if (pm > -INFINITY) s += ssum[(size_t)row * stat_ld + i] * __expf(pm - mx);
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float l = mx + __logf(s);
    if (lse) lse[row] = l;
    if (out_lp) out_lp[row] = tval[row] - l;
  }
}

}  // namespace

extern "C" int pgca_gemm_bf16(const pgca_gemm_args* args, void* stream) {
  const pgca_gemm_args& a = *args;
  if (!a.A || !a.B || a.M <= 0 || a.N <= 0 || a.K <= 0) {
    set_error("pgca_gemm_bf16: null operand or empty shape (M=%d N=%d K=%d)", a.M, a.N, a.K);
    return PGCA_ERR_INVALID;
  }
  if ((a.lda & 7) || (a.ldb & 7) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) {
    set_error("pgca_gemm_bf16: lda/ldb must be multiples of 8 and operands 16-B aligned (lda=%d ldb=%d)", a.lda,
              a.ldb);
    return PGCA_ERR_INVALID;
  }
  const bool a_kcontig = a.layout != PGCA_TN, b_kcontig = a.layout == PGCA_NT;
  if ((a_kcontig || b_kcontig) && (a.K & 7)) {
    set_error("pgca_gemm_bf16: K=%d must be a multiple of 8 for K-contiguous operands", a.K);
    return PGCA_ERR_INVALID;
  }
  if (a_kcontig ? a.lda < a.K : a.lda < ((a.M + 7) & ~7)) {
    set_error("pgca_gemm_bf16: lda=%d too small", a.lda);
    return PGCA_ERR_INVALID;
  }
  if (b_kcontig ? a.ldb < a.K : a.ldb < ((a.N + 7) & ~7)) {
    set_error("pgca_gemm_bf16: ldb=%d too small", a.ldb);
    return PGCA_ERR_INVALID;
  }
  if (a.epilogue == PGCA_EPI_ROWSTATS) {
    if (!a.stat_max || !a.stat_sum || a.stat_ld < 2 * ((a.N + BN - 1) / BN)) {
      set_error("pgca_gemm_bf16: ROWSTATS needs stat buffers with stat_ld >= %d", 2 * ((a.N + BN - 1) / BN));
      return PGCA_ERR_INVALID;
    }
  } else if (a.epilogue == PGCA_EPI_DLOGITS) {
    if (!a.row_lse || !a.row_scale || !a.targets || !a.out_bf16 || a.out_cols < a.N || a.ld_out_bf16 < a.out_cols) {
      set_error("pgca_gemm_bf16: DLOGITS needs row_lse,row_scale,targets,out_bf16 and out_cols >= N");
      return PGCA_ERR_INVALID;
    }
  } else {
    if (!a.out_bf16 && !a.out_f32) {
      set_error("pgca_gemm_bf16: no output buffer");
      return PGCA_ERR_INVALID;
    }
    if ((a.epilogue >= PGCA_EPI_DGELU_NEW && a.epilogue <= PGCA_EPI_DTANH) && !a.aux_in) {
      set_error("pgca_gemm_bf16: derivative epilogue needs aux_in");
      return PGCA_ERR_INVALID;
    }
  }
  const int ncols = a.epilogue == PGCA_EPI_DLOGITS ? a.out_cols : a.N;
  const int ntm = (a.M + BM - 1) / BM, ntn = (ncols + BN - 1) / BN;
  dim3 grid(ntm * ntn), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (a.layout) {
    case PGCA_NT: hipLaunchKernelGGL((gemm_kernel<0, 0>), grid, block, 0, s, a, ntm, ntn); break;
    case PGCA_NN: hipLaunchKernelGGL((gemm_kernel<0, 1>), grid, block, 0, s, a, ntm, ntn); break;
    case PGCA_TN: hipLaunchKernelGGL((gemm_kernel<1, 1>), grid, block, 0, s, a, ntm, ntn); break;
    default: set_error("pgca_gemm_bf16: unknown layout %d", a.layout); return PGCA_ERR_INVALID;
  }
  return check_launch("pgca_gemm_bf16");
}

extern "C" int pgca_rowstats_combine(const float* stat_max, const float* stat_sum, int32_t stat_ld, int32_t nparts,
                                     const float* target_val, int32_t M, float* lse, float* out_logprob,
                                     void* stream) {
  if (!stat_max || !stat_sum || M <= 0 || nparts <= 0 || (out_logprob && !target_val)) {
    set_error("pgca_rowstats_combine: bad arguments");
    return PGCA_ERR_INVALID;
  }
  hipLaunchKernelGGL(rowstats_combine_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, stat_max,
                     stat_sum, stat_ld, nparts, target_val, M, lse, out_logprob);
  return check_launch("pgca_rowstats_combine");
}
