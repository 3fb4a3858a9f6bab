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
#include <atomic>
#include <string>

#include "gemm_device.h"

namespace {

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
  // ... and inside that run tiles are visited in 8-row groups, N fastest within a group: the ~64 blocks an
  // XCD holds at once cover an 8 x 8 patch of tiles, so each A and B panel is fetched into its L2 once per 8 uses.
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * ntn;
  const int group = bid / per_group;
  const int first_m = group * GROUP_M;
  const int gsize = min(GROUP_M, ntm - first_m);
  const int in_group = bid - group * per_group;
  const int tm = first_m + in_group % gsize, tn = in_group / gsize;
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
  unsigned char* sbase = &smem[0][0][0];
  switch (a.epilogue) {
    case PGCA_EPI_ROWSTATS: epilogue_rowstats(a, acc, m0, n0, tn, wm, wn, lane); break;
    case PGCA_EPI_GELU_NEW: epilogue_store<PGCA_EPI_GELU_NEW>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_QUICK_GELU: epilogue_store<PGCA_EPI_QUICK_GELU>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_RELU: epilogue_store<PGCA_EPI_RELU>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_TANH: epilogue_store<PGCA_EPI_TANH>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_DGELU_NEW: epilogue_store<PGCA_EPI_DGELU_NEW>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_DRELU: epilogue_store<PGCA_EPI_DRELU>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_DTANH: epilogue_store<PGCA_EPI_DTANH>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_DLOGITS: epilogue_store<PGCA_EPI_DLOGITS>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_DQUICK_GELU: epilogue_store<PGCA_EPI_DQUICK_GELU>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_GELU_NEW_D: epilogue_store<PGCA_EPI_GELU_NEW_D>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    case PGCA_EPI_MUL_AUX: epilogue_store<PGCA_EPI_MUL_AUX>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
    default: epilogue_store<PGCA_EPI_NONE>(a, acc, sbase, m0, n0, wm, wn, lane, wave); break;
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


// ================================================================================================
// 256 x 256 x 64 tile, 8 waves (2 x 4, each 128 x 64), operands brought in by LDS-DMA
// (buffer_load ... lds, 16 B per lane, 1 KiB per wave-instruction) into a double-buffered 128 KiB
// LDS image: no VGPR staging and no ds_write traffic - on gfx950 the VGPR->LDS write path
// (~79 B/clk/CU for ds_write_b128) plus the fragment reads saturate the LDS port at the 128^2 tile.
// The LDS image of each DMA instruction is lane-linear, so the XOR swizzles are applied to the
// SOURCE address (same involution as the fragment readers).  Edges: M/N by clamping the source
// row/column (clamped rows only feed outputs the epilogue masks), K must be a multiple of 64.
// ================================================================================================
// RAW: bid_in is already the tile (tm * ntn + tn) - the grouped launch places its tiles itself.
template <int LA, int LB, bool RAW = false>
__device__ __forceinline__ void gemm256_body(const pgca_gemm_args& a, int ntm, int ntn, int nk_per_split, int bid_in,
                                             int ksplit, unsigned char* smem2) {  // smem2: [2 stages][A | B][32 KiB]
  const int nwg = ntm * ntn;
  int bid = bid_in;
  int tm, tn;
  if (RAW) {
    tm = bid / ntn;
    tn = bid - tm * ntn;
  } else {
    {
      const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * ntn;
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsize = min(GROUP_M, ntm - first_m);
    const int in_group = bid - group * per_group;
    tm = first_m + in_group % gsize;
    tn = in_group / gsize;
  }
  const int m0 = tm * BM2, n0 = tn * BN2;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B);
  const int nrows_b = a.epilogue == PGCA_EPI_DLOGITS ? a.N : a.N;

  Dma<LA> da;
  Dma<LB> db;
  da.init(lane, wave, a.lda, m0, LA == 0 ? a.M : ((a.M + 7) & ~7));
  db.init(lane, wave, a.ldb, n0, LB == 0 ? nrows_b : ((nrows_b + 7) & ~7));
  const bf16_t* abase = LA == 0 ? A + (size_t)m0 * a.lda : A + m0;
  const bf16_t* bbase = LB == 0 ? B + (size_t)n0 * a.ldb : B + n0;
  const size_t astep = LA == 0 ? (size_t)BK : (size_t)BK * a.lda;
  const size_t bstep = LB == 0 ? (size_t)BK : (size_t)BK * a.ldb;

  f32x4 acc[2][4][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K: blockIdx.y owns K tiles [kt0, kt0 + nk) and adds its partial with f32 atomics (accumulate == 2)
  const int kt0 = ksplit * nk_per_split;
  const int nk = min(nk_per_split, a.K / BK - kt0);
  abase += (size_t)kt0 * astep;
  bbase += (size_t)kt0 * bstep;
#ifdef PGCA_GEMM_TIMING
  unsigned long long ts0, ts1, ts2, ts3;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts0)::"memory");
#endif
  da.issue(abase, smem2, wave);
  db.issue(bbase, smem2 + TILE2_BYTES, wave);
  dma_wait();
  __syncthreads();
#ifdef PGCA_GEMM_TIMING
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1)::"memory");
#endif

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      unsigned char* nxt = smem2 + (cur ^ 1) * 2 * TILE2_BYTES;
      da.issue(abase + (size_t)(kt + 1) * astep, nxt, wave);
      db.issue(bbase + (size_t)(kt + 1) * bstep, nxt + TILE2_BYTES, wave);
    }
    const unsigned char* la = smem2 + cur * 2 * TILE2_BYTES;
    const unsigned char* lb = la + TILE2_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<LB, 512>(lb, wn * 64, j, kk, lane);
      mma_half<LA>(la, wm * 128, kk, lane, fb, acc[0]);
      mma_half<LA>(la, wm * 128 + 64, kk, lane, fb, acc[1]);
    }
    dma_wait();       // tile kt+1 has landed (this wave's pieces) ...
    __syncthreads();  // ... and every wave is done reading tile kt
  }

#ifdef PGCA_GEMM_TIMING
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts2)::"memory");
#endif
  run_epilogue(a, acc[0], smem2, m0 + wm * 128, n0, tn, wn, lane, wave);
  run_epilogue(a, acc[1], smem2, m0 + wm * 128 + 64, n0, tn, wn, lane, wave);
#ifdef PGCA_GEMM_TIMING
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts3)::"memory");
  if (a.stat_max && a.epilogue != PGCA_EPI_ROWSTATS && lane == 0) {
    float* o = a.stat_max + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = (float)(ts1 - ts0); o[1] = (float)(ts2 - ts1); o[2] = (float)(ts3 - ts2); o[3] = (float)nk;
  }
#endif
}

template <int LA, int LB>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const pgca_gemm_args a, int ntm, int ntn, int nk_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  gemm256_body<LA, LB>(a, ntm, ntn, nk_per_split, blockIdx.x, blockIdx.y, smem2);
}

// Grouped launch of up to four TN (weight-gradient) problems: C_p (+)= A_p^t B_p.  One decoder layer has four weight
// gradients with 16-64 output tiles each; launched one by one each needs split-K to fill 256 CUs and then pays
// 4 x M x N f32 atomics (memory-side, 1.3 TB/s chip-wide: ~48 us per launch).  Together they are 192 tiles with the
// whole K each: no split, no atomics, a plain read-modify-write epilogue.
struct pgca_group_param {
  pgca_gemm_args a[4];
  int start[5];  // first workgroup of problem p; start[count..4] = total
  int ntm[4], ntn[4];
};

__global__ __launch_bounds__(512, 2) void gemm256_group_tn_kernel(const pgca_group_param gp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  // Tile placement.  Workgroup b runs on XCD b % 8, and every XCD has its own 4 MiB L2: the tiles are laid out problem
  // after problem with the SHORTER tile dimension running fastest, and XCD x takes a contiguous run of that order - a
  // compact rectangle of (mostly) one problem, so the 24 tiles an XCD holds for a GPT-2-M block stream ~11 operand panels
  // through its L2 instead of ~24 when every problem is spread over all eight XCDs.
  int b;
  {
    const int total = gp.start[4];
    const int xcd = blockIdx.x & 7, q = total >> 3, r = total & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  int p = 0;
  if (b >= gp.start[1]) p = 1;
  if (b >= gp.start[2]) p = 2;
  if (b >= gp.start[3]) p = 3;
  // field-by-field selection with constant indices keeps everything in SGPRs (a dynamically indexed by-value
  // aggregate would be spilled to scratch and come back in VGPRs, which the LDS-DMA asm cannot take)
#define PGCA_SEL(F) (p == 0 ? gp.a[0].F : p == 1 ? gp.a[1].F : p == 2 ? gp.a[2].F : gp.a[3].F)
  pgca_gemm_args a = {};
  a.A = PGCA_SEL(A);
  a.B = PGCA_SEL(B);
  a.M = PGCA_SEL(M);
  a.N = PGCA_SEL(N);
  a.K = PGCA_SEL(K);
  a.lda = PGCA_SEL(lda);
  a.ldb = PGCA_SEL(ldb);
  a.out_f32 = PGCA_SEL(out_f32);
  a.ld_out_f32 = PGCA_SEL(ld_out_f32);
  a.alpha = PGCA_SEL(alpha);
  a.accumulate = PGCA_SEL(accumulate);
  a.layout = PGCA_TN;
  a.epilogue = PGCA_EPI_NONE;
#undef PGCA_SEL
  const int ntm = p == 0 ? gp.ntm[0] : p == 1 ? gp.ntm[1] : p == 2 ? gp.ntm[2] : gp.ntm[3];
  const int ntn = p == 0 ? gp.ntn[0] : p == 1 ? gp.ntn[1] : p == 2 ? gp.ntn[2] : gp.ntn[3];
  const int st = p == 0 ? gp.start[0] : p == 1 ? gp.start[1] : p == 2 ? gp.start[2] : gp.start[3];
  const int l = b - st;
  int tm, tn;
  if (ntm <= ntn) {
    tn = l / ntm;
    tm = l - tn * ntm;
  } else {
    tm = l / ntn;
    tn = l - tm * ntn;
  }
  gemm256_body<1, 1, true>(a, ntm, ntn, a.K / BK, tm * ntn + tn, 0, smem2);
}

constexpr size_t GEMM256_LDS = 4 * TILE2_BYTES;  // 128 KiB

int ensure_gemm256_attr() {
  static std::atomic<int> done{0};
  if (!done.load(std::memory_order_acquire)) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm256_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GEMM256_LDS);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm256_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GEMM256_LDS);
    hipError_t e3 = hipFuncSetAttribute((const void*)gemm256_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GEMM256_LDS);
    hipError_t e4 = hipFuncSetAttribute((const void*)gemm256_group_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GEMM256_LDS);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      (void)hipGetLastError();
      set_error("gemm256: cannot raise dynamic LDS limit");
      return PGCA_ERR_LAUNCH;
    }
    done.store(1, std::memory_order_release);  // idempotent: a racing second caller only repeats the same calls
  }
  return PGCA_OK;
}

// Kernel choice: big, K-aligned problems take the 256^2 LDS-DMA kernel, everything else the general 128^2
// one.  Gradient-accumulating GEMMs with few output tiles but a long K (weight gradients: K = tokens) are split
// along K so the 256^2 kernel still fills the chip; partials meet through f32 atomics.
int plan_tile(const pgca_gemm_args& a, int* splits_out) {
  const int ncols = a.epilogue == PGCA_EPI_DLOGITS ? a.out_cols : a.N;
  const int ntm2 = (a.M + BM2 - 1) / BM2, ntn2 = (ncols + BN2 - 1) / BN2;
  const int force = gemm_tuning().tile;
  const int nk_total = a.K / BK;
  int splits = 1;
  const bool splittable = a.epilogue == PGCA_EPI_NONE && a.accumulate && a.out_f32 && !a.out_bf16 && !a.bias &&
                          !a.residual;
  if (splittable && ntm2 * ntn2 < 192) {
    splits = 256 / (ntm2 * ntn2);
    if (splits > nk_total / 8) splits = nk_total / 8;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
  }
  const bool want256 = force ? force == 256 : (ntm2 * ntn2 * splits >= 192);
  *splits_out = splits;
  return ((a.K % BK) == 0 && want256 && a.M >= 8 && ncols >= 8) ? 256 : 128;
}

}  // namespace

// Schedule of the 256^2 tile: 0 = gemm256_kernel (2-stage BK=64 loop), 6 = gemm256s_kernel (gemm_phase.hip: phase-staggered
// wave groups, 4-stage BK=32 ring).  Schedules that did not pay (BK=32 ring with plain barriers, phased halves, a 4-wave
// 128x128-per-wave asm pipeline, a 256x128 two-workgroups-per-CU tile) were measured, documented in DESIGN.md section 5 and
// removed from the library.
static int plan_variant(const pgca_gemm_args& a) {
  const int forced = gemm_tuning().schedule;
  if (forced >= 0) return forced == 6 ? 6 : 0;
  // K-contiguous A (forward, data gradients, LM head): the phase-staggered loop is 3-12 % faster; the K-strided
  // weight-gradient layout pays two transposed LDS reads per fragment in the load phases and stays on the 2-stage loop.
  return a.layout == PGCA_TN ? 0 : 6;
}

namespace pgca {
GemmTuning& gemm_tuning() {
  static GemmTuning t = [] {
    auto env = [](const char* name, int dflt) {
      const char* e = getenv(name);
      return e ? atoi(e) : dflt;
    };
    GemmTuning v;
    v.tile = env("PGCA_GEMM_TILE", 0);
    v.schedule = env("PGCA_GEMM_RING", -1);
    v.group = env("PGCA_GEMM_NO_GROUP", 0) ? 0 : 1;
    v.stagger = env("PGCA_GEMM_STAGGER", 0);
    return v;
  }();
  return t;
}
}  // namespace pgca

extern "C" int pgca_set_option(const char* name, int32_t value) {
  if (!name) {
    set_error("pgca_set_option: null name");
    return PGCA_ERR_INVALID;
  }
  GemmTuning& t = gemm_tuning();
  const std::string n(name);
  if (n == "gemm_tile" && (value == 0 || value == 128 || value == 256)) t.tile = value;
  else if (n == "gemm_schedule" && (value == -1 || value == 0 || value == 6)) t.schedule = value;
  else if (n == "gemm_group" && (value == 0 || value == 1)) t.group = value;
  else if (n == "gemm_stagger" && value >= 0 && value <= 64) t.stagger = value;
  else {
    set_error("pgca_set_option: unknown option or value out of range (%s = %d)", name, value);
    return PGCA_ERR_INVALID;
  }
  return PGCA_OK;
}

extern "C" int pgca_gemm_plan(const pgca_gemm_args* args) {
  int splits = 1;
  const int tile = plan_tile(*args, &splits);
  if (tile != 256) return 12801;
  return plan_variant(*args) * 1000000 + 25600 + splits;
}

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
    if (((a.epilogue >= PGCA_EPI_DGELU_NEW && a.epilogue <= PGCA_EPI_DTANH) || a.epilogue == PGCA_EPI_DQUICK_GELU ||
         a.epilogue == PGCA_EPI_MUL_AUX) &&
        !a.aux_in) {
      set_error("pgca_gemm_bf16: derivative epilogue needs aux_in");
      return PGCA_ERR_INVALID;
    }
  }
  if (a.colsum_part && ((a.epilogue != PGCA_EPI_DGELU_NEW && a.epilogue != PGCA_EPI_MUL_AUX) || a.ld_colsum < a.N ||
                        a.accumulate == 2)) {
    set_error("pgca_gemm_bf16: colsum_part needs the DGELU_NEW / MUL_AUX epilogue, ld_colsum >= N and no split-K");
    return PGCA_ERR_INVALID;
  }
  const int ncols = a.epilogue == PGCA_EPI_DLOGITS ? a.out_cols : a.N;
  hipStream_t s = (hipStream_t)stream;
  {
    const int ntm2 = (a.M + BM2 - 1) / BM2, ntn2 = (ncols + BN2 - 1) / BN2;
    const int nk_total = a.K / BK;
    int splits = 1;
    const bool want256 = plan_tile(a, &splits) == 256;
    if (want256) {
      if (ensure_gemm256_attr()) return PGCA_ERR_LAUNCH;
      pgca_gemm_args b = a;
      if (splits > 1) b.accumulate = 2;
      const int nkps = (nk_total + splits - 1) / splits;
      dim3 grid2(ntm2 * ntn2, (nk_total + nkps - 1) / nkps), block2(512);
      const int variant = plan_variant(a);
      if (variant == 6) return launch_gemm256s(b, ntm2, ntn2, nkps, (int)grid2.y, stream);
      switch (a.layout) {
        case PGCA_NT: hipLaunchKernelGGL((gemm256_kernel<0, 0>), grid2, block2, GEMM256_LDS, s, b, ntm2, ntn2, nkps); break;
        case PGCA_NN: hipLaunchKernelGGL((gemm256_kernel<0, 1>), grid2, block2, GEMM256_LDS, s, b, ntm2, ntn2, nkps); break;
        case PGCA_TN: hipLaunchKernelGGL((gemm256_kernel<1, 1>), grid2, block2, GEMM256_LDS, s, b, ntm2, ntn2, nkps); break;
        default: set_error("pgca_gemm_bf16: unknown layout %d", a.layout); return PGCA_ERR_INVALID;
      }
      return check_launch("pgca_gemm_bf16(256)");
    }
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = (ncols + BN - 1) / BN;
  dim3 grid(ntm * ntn), block(256);
  switch (a.layout) {
    case PGCA_NT: hipLaunchKernelGGL((gemm_kernel<0, 0>), grid, block, 0, s, a, ntm, ntn); break;
    case PGCA_NN: hipLaunchKernelGGL((gemm_kernel<0, 1>), grid, block, 0, s, a, ntm, ntn); break;
    case PGCA_TN: hipLaunchKernelGGL((gemm_kernel<1, 1>), grid, block, 0, s, a, ntm, ntn); break;
    default: set_error("pgca_gemm_bf16: unknown layout %d", a.layout); return PGCA_ERR_INVALID;
  }
  return check_launch("pgca_gemm_bf16");
}

extern "C" int pgca_gemm_bf16_grouped(const pgca_gemm_args* args, int32_t count, void* stream) {
  if (!args || count <= 0) {
    set_error("pgca_gemm_bf16_grouped: no problems");
    return PGCA_ERR_INVALID;
  }
  bool groupable = count <= 4 && gemm_tuning().group;
  for (int i = 0; i < count && groupable; ++i) {
    const pgca_gemm_args& a = args[i];
    groupable = a.layout == PGCA_TN && a.A && a.B && a.K > 0 && (a.K % BK) == 0 && a.M >= 8 && a.N >= 8 &&
                a.epilogue == PGCA_EPI_NONE && a.out_f32 && !a.out_bf16 && !a.bias && !a.residual &&
                !a.drop_threshold && !(a.lda & 7) && !(a.ldb & 7) && !((uintptr_t)a.A & 15) && !((uintptr_t)a.B & 15) &&
                a.lda >= ((a.M + 7) & ~7) && a.ldb >= ((a.N + 7) & ~7);
  }
  if (!groupable) {  // anything else: the ordinary path, one launch per problem
    for (int i = 0; i < count; ++i) {
      const int rc = pgca_gemm_bf16(&args[i], stream);
      if (rc) return rc;
    }
    return PGCA_OK;
  }
  if (ensure_gemm256_attr()) return PGCA_ERR_LAUNCH;
  pgca_group_param gp;
  int total = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < count) {
      gp.a[i] = args[i];
      gp.ntm[i] = (args[i].M + BM2 - 1) / BM2;
      gp.ntn[i] = (args[i].N + BN2 - 1) / BN2;
      gp.start[i] = total;
      total += gp.ntm[i] * gp.ntn[i];
    } else {
      gp.a[i] = args[count - 1];
      gp.ntm[i] = gp.ntn[i] = 1;
      gp.start[i] = 0x7fffffff;
    }
  }
  gp.start[4] = total;
  hipLaunchKernelGGL(gemm256_group_tn_kernel, dim3(total), dim3(512), GEMM256_LDS, (hipStream_t)stream, gp);
  return check_launch("pgca_gemm_bf16_grouped");
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
