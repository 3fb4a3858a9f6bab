// Wide-wave variant of the 256 x 256 LDS-DMA GEMM (gfx950): FOUR waves (2 x 2), one per SIMD, each owning a
// 128 x 128 block = 64 MFMA 16x16x32 accumulators = 256 AGPRs.  Why: with eight waves of 128 x 64 the fragment
// reads (192 KiB per 64-deep K tile per CU) plus the 64 KiB the LDS-DMA writes keep the LDS port busy for all
// 2048 clocks the tile's MFMAs take, so every hiccup is MFMA idle time.  128 x 128 per wave needs 128 KiB of
// fragment reads: the port is busy ~75 % of the MFMA time.
//
// One wave per SIMD means nobody else hides this wave's latencies, so the main loop is an explicit software
// pipeline written as ordered asm statements (the compiler neither reorders nor waits on them):
//   * K advances in 32-deep ring tiles, FOUR LDS stages (A 16 KiB x 4 | B 16 KiB x 4), LDS-DMA three tiles ahead,
//     retired with counted s_waitcnt vmcnt;
//   * the fragments of tile t+1 are read (ds_read_b128 / ds_read_b64_tr_b16) in between the 64 MFMAs of tile t,
//     two register sets alternating; one s_waitcnt lgkmcnt(0) per tile, when they are long back;
//   * accumulators are pinned to AGPRs by the asm constraint (left alone, hipcc picks the VGPR form of the MFMA
//     here and shuffles 400 v_accvgpr moves per tile between the register files).
#include "gemm_device.h"

// timing experiments only (wrong results): drop the DMA or the fragment reads from the MFMA stream
#ifndef PGCA_W_NODMA
#define PGCA_W_NODMA 0
#endif
#ifndef PGCA_W_NOREAD
#define PGCA_W_NOREAD 0
#endif

namespace {

constexpr int WBK = 32;
constexpr int WTILE = 256 * WBK * 2;   // 16 KiB per operand per stage
constexpr int WSTAGES = 4;
constexpr int WB_OFF = WSTAGES * WTILE;  // B stages start at 64 KiB

__device__ __forceinline__ int swz4w(int q) { return (0x78 >> (2 * q)) & 3; }  // {0,2,3,1}

// LDS-DMA of one operand tile (16 pieces of 1 KiB), 4 pieces per wave.  Images as in gemm256r_kernel.
template <int KS>
struct DmaW {
  unsigned goff[4];
  __device__ __forceinline__ void init(int lane, int wave, int ld, int origin, int extent) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wave * 4 + i;
      if (KS == 0) {  // [256 rows][32 k]: piece = 16 rows x 64 B; chunk c of row r at c ^ swz4((r>>2)&3)
        const int r = 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ swz4w((lane >> 4) & 3);
        const int rg = min(origin + r, extent - 1) - origin;
        goff[i] = (unsigned)(rg * ld + c * 8) * 2u + 3072u - 1024u * i;
      } else {        // [32 k][256 cols]: piece = 2 k-rows x 512 B
        const int k = 2 * j + (lane >> 5);
        const int c16 = lane & 31;
        const int h = (k & 3) | (((k >> 3) & 1) << 2);
        const int col = (((c16 >> 1) ^ h) << 4) + ((c16 & 1) << 3);
        const int cg = min(origin + col, extent - 8) - origin;
        goff[i] = (unsigned)(k * ld + cg) * 2u + 3072u - 1024u * i;
      }
    }
  }
  // All four pieces of a wave go through ONE M0 value: the instruction's immediate offset (I KiB) advances the LDS
  // address AND the memory address, so the per-piece voffset carries (goff - I KiB) and the resource base sits
  // 3 KiB low to keep every voffset non-negative.  Rewriting M0 between back-to-back LDS-DMA instructions
  // serialises them (~45 clocks of blocked issue each, s_memtime measurement).
  static __device__ __forceinline__ u32x4 rsrc(const bf16_t* base) {
    const unsigned long long b = (unsigned long long)base - 3072ull;
    u32x4 rs;
    rs[0] = (unsigned)b;
    rs[1] = (unsigned)(b >> 32) & 0xffffu;
    rs[2] = 0x7ffffff0u;
    rs[3] = 0x00020000u;
    return rs;
  }
  template <int I>
  __device__ __forceinline__ void piece(const u32x4& rs, unsigned lds_wave) const {
    if (I == 0) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :
                   : "s"(lds_wave), "v"(goff[0]), "s"(rs)
                   : "memory");
    } else {
      asm volatile("buffer_load_dwordx4 %0, %1, 0 offen offset:%2 lds" : : "v"(goff[I]), "s"(rs), "n"(I * 1024) : "memory");
    }
  }
  __device__ __forceinline__ void issue(const bf16_t* base, unsigned lds_tile, int wave) const {
    const u32x4 rs = rsrc(base);
    const unsigned lw = lds_tile + (unsigned)wave * 4096u;
    piece<0>(rs, lw);
    piece<1>(rs, lw);
    piece<2>(rs, lw);
    piece<3>(rs, lw);
  }
};

// Per-lane LDS byte addresses (stage 0) of the wave's eight 16-wide fragments of one operand.
template <int KS>
struct FragAddr {
  unsigned addr[KS == 0 ? 1 : 8];
  __device__ __forceinline__ void init(unsigned lds_operand, int wbase, int lane) {
    if (KS == 0) {
      const int row = wbase + (lane & 15);
      const int pos = (lane >> 4) ^ swz4w((lane >> 2) & 3);
      addr[0] = lds_operand + (unsigned)(row * 64 + pos * 16);  // fragment `sub` is 1 KiB further
    } else {
      const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
      const int k = 8 * g + q;
      const int h = q | ((g & 1) << 2);
#pragma unroll
      for (int sub = 0; sub < 8; ++sub) {
        const int s32 = (wbase >> 4) + sub;
        addr[KS == 0 ? 0 : sub] = lds_operand + (unsigned)(k * 512 + ((s32 ^ h) << 5) + 8 * p);  // rows k+4: +2 KiB
      }
    }
  }
};

template <int KS, int SUB>
__device__ __forceinline__ void read_frag_w(u32x4& f, const FragAddr<KS>& fa, unsigned stage_off) {
  if (KS == 0) {
    const unsigned a = fa.addr[0] + stage_off;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"(SUB * 1024));
  } else {
    const unsigned a = fa.addr[KS == 0 ? 0 : SUB] + stage_off;
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(a));
    f = (u32x4){lo[0], lo[1], hi[0], hi[1]};
  }
}

__device__ __forceinline__ void mfma_agpr(f32x4& c, const u32x4& a, const u32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One ring tile: 64 MFMAs on (fa, fb); the fragments of the NEXT tile are fetched into (na, nb) in between.
// READ_NEXT is a compile-time flag so that the drained tail of the loop carries no dead reads.
// The eight LDS-DMA instructions that fetch tile t+3 go out one per row of MFMAs as well (issued back to back
// they cost the wave ~430 clocks per tile with the matrix pipe idle - s_memtime measurement).
template <int LA, int LB, bool READ_NEXT>
__device__ __forceinline__ void tile_step(f32x4 (&acc)[2][2][4][4], const u32x4 (&fa)[8], const u32x4 (&fb)[8],
                                          u32x4 (&na)[8], u32x4 (&nb)[8], const FragAddr<LA>& aa,
                                          const FragAddr<LB>& ab, unsigned next_stage_off, bool dma,
                                          const DmaW<LA>& da, const DmaW<LB>& db, const u32x4& rsa, const u32x4& rsb,
                                          unsigned lwa, unsigned lwb) {
#define PGCA_W_ROW(I)                                                     \
  {                                                                       \
    if (dma && !PGCA_W_NODMA) {                                           \
      if ((I) < 4) da.template piece<(I) & 3>(rsa, lwa);                  \
      else db.template piece<(I) & 3>(rsb, lwb);                          \
    }                                                                     \
    if (READ_NEXT && !PGCA_W_NOREAD) read_frag_w<LB, I>(nb[I], ab, next_stage_off); \
    mfma_agpr(acc[(I) >> 2][0][(I) & 3][0], fa[I], fb[0]);                \
    mfma_agpr(acc[(I) >> 2][0][(I) & 3][1], fa[I], fb[1]);                \
    mfma_agpr(acc[(I) >> 2][0][(I) & 3][2], fa[I], fb[2]);                \
    mfma_agpr(acc[(I) >> 2][0][(I) & 3][3], fa[I], fb[3]);                \
    if (READ_NEXT && !PGCA_W_NOREAD) read_frag_w<LA, I>(na[I], aa, next_stage_off); \
    mfma_agpr(acc[(I) >> 2][1][(I) & 3][0], fa[I], fb[4]);                \
    mfma_agpr(acc[(I) >> 2][1][(I) & 3][1], fa[I], fb[5]);                \
    mfma_agpr(acc[(I) >> 2][1][(I) & 3][2], fa[I], fb[6]);                \
    mfma_agpr(acc[(I) >> 2][1][(I) & 3][3], fa[I], fb[7]);                \
  }
  PGCA_W_ROW(0) PGCA_W_ROW(1) PGCA_W_ROW(2) PGCA_W_ROW(3) PGCA_W_ROW(4) PGCA_W_ROW(5) PGCA_W_ROW(6) PGCA_W_ROW(7)
#undef PGCA_W_ROW
}

#ifdef PGCA_GEMM_TIMING
#define TSTAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#else
#define TSTAMP(v)
#endif

template <int LA, int LB>
__global__ __launch_bounds__(256, 1) void gemm256w_kernel(const pgca_gemm_args a, int ntm, int ntn, int nk_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];  // [A x 4 stages | B x 4 stages], 16 KiB each

  const int nwg = ntm * ntn;
  int bid = blockIdx.x;
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
  const int tm = first_m + in_group % gsize, tn = in_group / gsize;
  const int m0 = tm * BM2, n0 = tn * BN2;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B);

  DmaW<LA> da;
  DmaW<LB> db;
  da.init(lane, wave, a.lda, m0, LA == 0 ? a.M : ((a.M + 7) & ~7));
  db.init(lane, wave, a.ldb, n0, LB == 0 ? a.N : ((a.N + 7) & ~7));
  const bf16_t* abase = LA == 0 ? A + (size_t)m0 * a.lda : A + m0;
  const bf16_t* bbase = LB == 0 ? B + (size_t)n0 * a.ldb : B + n0;
  const size_t astep = LA == 0 ? (size_t)WBK : (size_t)WBK * a.lda;
  const size_t bstep = LB == 0 ? (size_t)WBK : (size_t)WBK * a.ldb;

  const unsigned lds_a = (unsigned)(size_t)LDS_PTR(smemw), lds_b = lds_a + WB_OFF;
  FragAddr<LA> aa;
  FragAddr<LB> ab;
  aa.init(lds_a, wm * 128, lane);
  ab.init(lds_b, wn * 128, lane);

  f32x4 acc[2][2][4][4];  // [row half][col half] 64 x 64 blocks
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][g][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K in units of 64-deep tiles (two ring tiles each), like the 8-wave kernel; nk is even
  const int kt0 = blockIdx.y * nk_per_split * 2;
  const int nk = min(nk_per_split * 2, a.K / WBK - kt0);
  abase += (size_t)kt0 * astep;
  bbase += (size_t)kt0 * bstep;

#pragma unroll
  for (int p = 0; p < WSTAGES - 1; ++p) {
    if (p < nk) {
      da.issue(abase + (size_t)p * astep, lds_a + p * WTILE, wave);
      db.issue(bbase + (size_t)p * bstep, lds_b + p * WTILE, wave);
    }
  }
  // tile 0 landed (tiles 1, 2 may stay in flight: 8 DMA instructions per tile per wave)
  if (nk >= 3) wait_vm<16>();
  else wait_vm<0>();   // nk == 2: retiring both costs nothing that matters
  __builtin_amdgcn_s_barrier();

  u32x4 f0a[8], f0b[8], f1a[8], f1b[8];
  {
#define PGCA_W_RD0(I) read_frag_w<LB, I>(f0b[I], ab, 0u); read_frag_w<LA, I>(f0a[I], aa, 0u);
    PGCA_W_RD0(0) PGCA_W_RD0(1) PGCA_W_RD0(2) PGCA_W_RD0(3) PGCA_W_RD0(4) PGCA_W_RD0(5) PGCA_W_RD0(6) PGCA_W_RD0(7)
#undef PGCA_W_RD0
  }

#ifdef PGCA_GEMM_TIMING
  unsigned long long t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, tstart = 0;
  unsigned long long d_mma = 0, d_bar = 0, d_dma = 0, d_lgk = 0, d_vm = 0, t4p = 0;
  TSTAMP(tstart);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  // Iteration kt: [tile kt+1 visible] -> issue DMA(kt+3) -> 64 MFMAs on tile kt's fragments, reading tile kt+1's.
  for (int kt = 0; kt < nk; kt += 2) {
    // ---- even tile: compute from f0, fetch f1 (tile kt+1 always exists: nk is even)
    if (kt + 2 < nk) wait_vm<8>();   // tile kt+1 landed, tile kt+2 may stay in flight
    else wait_vm<0>();
    TSTAMP(t1);
    __builtin_amdgcn_s_barrier();
    TSTAMP(t2);
    const unsigned st3 = (unsigned)((kt + 3) & 3) * WTILE + (unsigned)wave * 4096u;
    const u32x4 rsa3 = DmaW<LA>::rsrc(abase + (size_t)(kt + 3) * astep), rsb3 = DmaW<LB>::rsrc(bbase + (size_t)(kt + 3) * bstep);
    TSTAMP(t3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef PGCA_GEMM_TIMING
    TSTAMP(t4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (kt > 0) { d_mma += t5 - t4p; d_vm += t1 - t5; }
    d_bar += t2 - t1; d_dma += t3 - t2; d_lgk += t4 - t3; t4p = t4;
#endif
    tile_step<LA, LB, true>(acc, f0a, f0b, f1a, f1b, aa, ab, (unsigned)((kt + 1) & 3) * WTILE, kt + 3 < nk, da, db, rsa3,
                            rsb3, lds_a + st3, lds_b + st3);
    TSTAMP(t5);

    // ---- odd tile: compute from f1, fetch f0 (tile kt+2; past the end the reads hit a stale stage and are unused:
    // keeping the step branch-free keeps the 256 accumulators out of control-flow merges)
    if (kt + 3 < nk) wait_vm<8>();
    else wait_vm<0>();
    TSTAMP(t1);
    __builtin_amdgcn_s_barrier();
    TSTAMP(t2);
    const unsigned st4 = (unsigned)((kt + 4) & 3) * WTILE + (unsigned)wave * 4096u;
    const u32x4 rsa4 = DmaW<LA>::rsrc(abase + (size_t)(kt + 4) * astep), rsb4 = DmaW<LB>::rsrc(bbase + (size_t)(kt + 4) * bstep);
    TSTAMP(t3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef PGCA_GEMM_TIMING
    TSTAMP(t4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d_mma += t5 - t4p; d_vm += t1 - t5; d_bar += t2 - t1; d_dma += t3 - t2; d_lgk += t4 - t3; t4p = t4;
#endif
    tile_step<LA, LB, true>(acc, f1a, f1b, f0a, f0b, aa, ab, (unsigned)((kt + 2) & 3) * WTILE, kt + 4 < nk, da, db, rsa4,
                            rsb4, lds_a + st4, lds_b + st4);
    TSTAMP(t5);
  }
#ifdef PGCA_GEMM_TIMING
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (a.stat_max && lane == 0) {
    float* o = a.stat_max + ((size_t)blockIdx.x * 4 + wave) * 8;
    TSTAMP(t1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    o[0] = (float)d_mma; o[1] = (float)d_bar; o[2] = (float)d_dma; o[3] = (float)d_lgk; o[4] = (float)(t1 - tstart);
    o[5] = (float)nk; o[6] = (float)d_vm;
  }
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results (written from asm) settle before they are read
  __syncthreads();                                     // the epilogue stages through the same LDS

  run_epilogue_trunk(a, acc[0][0], smemw, m0 + wm * 128, n0, wn * 2, lane, wave);
  run_epilogue_trunk(a, acc[0][1], smemw, m0 + wm * 128, n0, wn * 2 + 1, lane, wave);
  run_epilogue_trunk(a, acc[1][0], smemw, m0 + wm * 128 + 64, n0, wn * 2, lane, wave);
  run_epilogue_trunk(a, acc[1][1], smemw, m0 + wm * 128 + 64, n0, wn * 2 + 1, lane, wave);
}

constexpr size_t GEMM256W_LDS = 2 * WSTAGES * WTILE;  // 128 KiB

template <int LA, int LB>
int launch_one(const pgca_gemm_args& a, int ntm, int ntn, int nkps, int nsplit, hipStream_t s) {
  static int attr_done = 0;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)gemm256w_kernel<LA, LB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)GEMM256W_LDS) != hipSuccess) {
      (void)hipGetLastError();
      set_error("gemm256w: cannot raise dynamic LDS limit");
      return PGCA_ERR_LAUNCH;
    }
    attr_done = 1;
  }
  hipLaunchKernelGGL((gemm256w_kernel<LA, LB>), dim3(ntm * ntn, nsplit), dim3(256), GEMM256W_LDS, s, a, ntm, ntn, nkps);
  return check_launch("pgca_gemm_bf16(256 wide-wave)");
}

}  // namespace

int pgca::launch_gemm256w(const pgca_gemm_args& a, int ntm, int ntn, int nk_per_split, int nsplit, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!trunk_epilogue(a.epilogue)) return 1;
  switch (a.layout) {
    case PGCA_NT: return launch_one<0, 0>(a, ntm, ntn, nk_per_split, nsplit, s);
    case PGCA_NN: return launch_one<0, 1>(a, ntm, ntn, nk_per_split, nsplit, s);
    case PGCA_TN: return launch_one<1, 1>(a, ntm, ntn, nk_per_split, nsplit, s);
    default: return 1;
  }
}
